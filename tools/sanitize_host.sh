#!/usr/bin/env bash
# Host-side sanitizer pass over libfiveeq_hip.so: AddressSanitizer + UBSan on the HOST code of the
# C ABI (argument validation, model preparation, the Latin-hypercube host twin, plan bookkeeping), run in the CPU
# container.  Device code is NOT instrumented (-fno-gpu-sanitize): GPU sanitizers are not available on this pool.
#   bash tools/sanitize_host.sh            # builds /tmp/libfiveeq_hip_asan.so and runs the CPU C-ABI tests against it
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/libfiveeq_hip_asan.so
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
HASH=$(cat "$R/fiveeqscm_amd/csrc/fiveeq_capi.hip" "$R/fiveeqscm_amd/csrc/fiveeq_device.hpp" "$R/include/fiveeq.h" | sha256sum | cut -c1-64)
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared --offload-arch=gfx950 -I "$R/include" \
    -DFIVEEQ_SOURCE_HASH="\"$HASH\"" \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -shared-libsan \
    -o "$OUT" "$R/fiveeqscm_amd/csrc/fiveeq_capi.hip"
cd "$R"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    FIVEEQ_LIB_PATH="$OUT" python3 -m pytest -q -p no:cacheprovider tests/test_capi_cpu.py tests/test_lhs.py
# ThreadSanitizer over the same host code, for the one test that calls the C ABI from nine threads at once (the header's
# concurrency contract: thread-local error text, the atomic fp32 packing switch).  Python itself is not instrumented: TSan
# sees the library's own accesses, which is where a data race of the library would be.
OUT_T=${TMPDIR:-/tmp}/libfiveeq_hip_tsan.so
RT_T=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.tsan-x86_64.so" | head -1)
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared --offload-arch=gfx950 -I "$R/include" \
    -DFIVEEQ_SOURCE_HASH="\"$HASH\"" -fsanitize=thread -fno-gpu-sanitize -shared-libsan -o "$OUT_T" "$R/fiveeqscm_amd/csrc/fiveeq_capi.hip"
LD_PRELOAD="$RT_T" TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 FIVEEQ_LIB_PATH="$OUT_T" \
    python3 -m pytest -q -p no:cacheprovider tests/test_capi_cpu.py -k "concurrent or validation"
