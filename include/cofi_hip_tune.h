/*
 * cofi_hip_tune.h — tuning / test hooks of libcofi_hip.so.  NOT part of the drop-in ABI (include/cofi_hip.h): nothing in
 * cofii2p_amd/ or model/ calls these; tools/ (plan sweeps, A/B probes) and tests/ (bit-equality of one contraction under
 * two tilings) do.
 *
 * Every hook sets an override of the CALLING THREAD only (thread_local in csrc/gemm.hip): plans are chosen on the thread
 * that enqueues a launch, so a tool or test forcing a plan cannot change what any other thread - a serving thread, a
 * loader thread - launches.  Overrides stay until reset by the value documented as "default".  The results of a forced
 * plan are those of the same contraction under another tiling / K split (the tests compare them bit for bit where the
 * summation order is equal); no hook changes numerics beyond that.
 */
#ifndef COFI_HIP_TUNE_H
#define COFI_HIP_TUNE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Register-staged kernels (cofi_gemm_f32* / cofi_conv2d_nhwc*): force the tile (bm, bn in {64, 128}; 128 x 64 exists for the
 * bf16x6 arithmetic only) and the K split of every following plan; ksplit 0 = as the plan would choose.  (0, 0, 0) = default
 * (plan tables + heuristic).  Also keeps the 256 x 128 and the direct 3 x 3 kernels off while a tile is forced. */
int cofi_tune_force_plan(int bm, int bn, int ksplit);

/* Pre-split-operand kernel (COFI_GEMM_A_SPLIT): cfg >= 0 = that row of the configuration table with `ksplit`; -2 = send
 * pre-split operands to the register-staged kernel instead; -1 = default. */
int cofi_tune_force_planes(int cfg, int ksplit);

/* 256 x 128 one-workgroup-per-CU kernels (bf16x6 and f16x3): mode 1 = on every eligible launch (ksplit 0 = the cost model's
 * split), -1 = never, 0 = default (table + cost model). */
int cofi_tune_force_big(int mode, int ksplit);

/* Direct 3 x 3 convolution (64 output channels): 1 = on every eligible convolution regardless of the tile count, 2 = ... with
 * 4-row tiles, -1 = never, 0 = default. */
int cofi_tune_force_conv_direct(int mode);

/* Bit 64: the generic row-wise epilogue instead of the straight-line one (identical bits, slower: the A/B of DESIGN 14.3).
 * Bit 256: the f16x3 kernel in its four-wave geometry (one wave per SIMD) instead of the eight-wave one (identical bits).
 * Bit 512: the f16x3 kernel with pre-split weights in its 256-row one-workgroup-per-CU form instead of 128 x 128 tiles, two per CU.
 * Other bits are unused.  0 = default. */
int cofi_tune_big_debug(int flags);

/* Diagnostic of the f16x3 kernel (COFI_GEMM_F16X3): K-tiles re-split because they left the fp16 window of their panel's running scale,
 * summed over all launches since the last reset (a device-side counter; the call synchronises the device).  reset != 0: zero it after
 * reading.  -1 on error. */
long cofi_tune_f16x3_resplit_events(int reset);

/* Host-side census: contractions the calling thread enqueued (or captured) on the f16x3 kernel since the last reset -> their count;
 * *flops (optional) = the sum of their 2 M N K.  bench.py prices a launch list that mixes the six-product and the three-product kernel
 * against the mix of their two matrix roofs with it.  reset != 0: zero both after reading. */
long cofi_tune_f16x3_launch_flops(int reset, double *flops);

#ifdef __cplusplus
}
#endif
#endif
