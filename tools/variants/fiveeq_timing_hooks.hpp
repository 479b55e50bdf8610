// fiveeq_timing_hooks.hpp — EXPERIMENT BUILDS ONLY (-DFIVEEQ_FUSED_TIMING; tools/fused_timing.py).  Not part of the product:
// fiveeq_device.hpp includes this file only under that flag, and a library built with it reports it in fiveeq_build_flags()
// (the shipped library's string is empty; tested).
//
// The hooks OVERWRITE result records with timestamps — a timing build's statistics / histograms are not results:
//   fused_kernel : the wave's statistics record of the LAST step becomes (start, end [100 MHz wall clock], HW_ID, XCC_ID)
#pragma once

#ifdef FIVEEQ_FUSED_TIMING
#define FIVEEQ_HOOK_FUSED_BEGIN const unsigned long long dbg_t0 = wall_clock64();
#define FIVEEQ_HOOK_FUSED_END                                                           \
    if (wave_live && (threadIdx.x & 63) == 0) {                                         \
        double* o = stats + ((int64_t)W * wave * n_steps + (t_end - 1)) * 4;            \
        o[0] = (double)dbg_t0;                                                          \
        o[1] = (double)wall_clock64();                                                  \
        o[2] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);                       \
        o[3] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);                      \
    }
#endif
