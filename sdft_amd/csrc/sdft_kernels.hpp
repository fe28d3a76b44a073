// sdft_kernels.hpp -- the hand-written HIP kernels for gfx950 (CDNA4) of the modulated Sliding DFT hot path, by stage.
// Citations in the stage files are into /root/reference/c/src/sdft/sdft.h.
//
//   sdft_base.hpp          complex helpers, cross-lane (DPP) neighbour fetch, K0 delta_kernel (differences + delay line),
//                          the recurrence step (sdft.h:572-585), the completion word of short synchronous calls
//   sdft_carry_fast.hpp    K1a/K1b  carries of the chunk-parallel FD double path: chunk_sum_kernel, chunk_fft_kernel,
//                          chunk_fft_mixed_kernel (partial sums per chunk), carry_scan_kernel (scan over chunks)
//   sdft_carry_exact.hpp   K1a'  the reference's rounding sequence (FD float always): carry_exact_kernel (serial pass),
//                          fid_seed_kernel + carry_relay_kernel (one dependent addition per step on the chain)
//   sdft_forward.hpp       K1    window (sdft.h:350-402), ForwardArgs, flow-mode waits, self-carried chunks (fold + FFT in
//                          LDS), forward_kernel (independent waves with halo lanes: any N, row-pointer outputs)
//   sdft_forward_hop.hpp   K1h   calls of one time chunk: forward_hop_kernel, forward_hop2_kernel (two waves per tile)
//   sdft_ops.hpp           spectral operations of the fused call, the synthesis term (sdft.h:641-651), user_rows_kernel
//   sdft_forward_rows.hpp  K1    forward_rows_kernel: one workgroup per (chunk, row), LDS edge exchange, lockstep row
//                          stores -- the dominant kernel; SYN != 0: fused synthesis on the windowed rows
//   sdft_fused.hpp         K3    fold_coeff_kernel, process_rows_kernel, process_hop2_kernel: analysis -> operation ->
//                          synthesis with window, operation and synthesis folded into one coefficient per bin
//   sdft_inverse.hpp       K2    inverse_exact_kernel (reference order, streaming), inverse_row_kernel (few rows),
//                          inverse_kernel (tree sum, with the rounding-interval proof of the reference's float), scale_rows_kernel
//
// Common decomposition: lanes <-> frequency bins (one complex bin per lane for 16-byte bins, two adjacent bins per lane
// for 8-byte bins, so a lane always stores 16 B), the sample loop is carried inside the kernel, the grid is
// bins x time chunks x channels.  The per-sample input difference is wave-uniform and arrives over the scalar unit
// (s_load through the constant address space); twiddles and state live in VGPRs for a whole chunk; window neighbours
// come from DPP whole-wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1).
//
// Arithmetic follows the reference's struct-complex formulas operation by operation and the translation units are
// compiled with -ffp-contract=off: given the same carry-in a wave reproduces the reference bit for bit (the FUSED
// instantiations of forward_rows_kernel and process_rows_kernel are the deliberate exceptions, used only where the
// carry-in already differs in summation order).
//
// This text is also compiled at run time, by hiprtc, for sdft_hip_process_n with sdft_hip_op_expr: build.py inlines the
// stage files into one string the library carries (the run-time compiler brings its own HIP declarations).

#pragma once

#include "sdft_base.hpp"
#include "sdft_carry_fast.hpp"
#include "sdft_carry_exact.hpp"
#include "sdft_forward.hpp"
#include "sdft_forward_hop.hpp"
#include "sdft_ops.hpp"
#include "sdft_forward_rows.hpp"
#include "sdft_fused.hpp"
#include "sdft_inverse.hpp"
