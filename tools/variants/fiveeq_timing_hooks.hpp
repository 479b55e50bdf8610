// fiveeq_timing_hooks.hpp — EXPERIMENT BUILDS ONLY (-DFIVEEQ_FUSED_TIMING / -DFIVEEQ_TILE_TIMING; tools/fused_timing.py,
// tools/tile_timing.py).  Not part of the product: fiveeq_device.hpp includes this file only under one of those flags, and
// a library built with either reports it in fiveeq_build_flags() (the shipped library's string is empty; tested).
//
// The hooks OVERWRITE result records with timestamps — a timing build's statistics / histograms are not results:
//   fused_kernel : the wave's statistics record of the LAST step becomes (start, end [100 MHz wall clock], HW_ID, XCC_ID)
//   tile_kernel  : the histogram counters of the tile's LAST step row become, per workgroup, the same four words
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

#ifdef FIVEEQ_TILE_TIMING
#define FIVEEQ_HOOK_TILE_BEGIN const unsigned long long dbg_t0 = wall_clock64();
#define FIVEEQ_HOOK_TILE_END                                                            \
    if (do_hist) {                                                                      \
        __syncthreads();                                                                \
        if (threadIdx.x == 0 && (blockIdx.x + 1) * 4 <= n_bins) {                       \
            unsigned long long* o = hist + (int64_t)(t_end - 1) * n_bins + blockIdx.x * 4; \
            o[0] = dbg_t0;                                                              \
            o[1] = wall_clock64();                                                      \
            o[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                           \
            o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                          \
        }                                                                               \
    }
#endif
