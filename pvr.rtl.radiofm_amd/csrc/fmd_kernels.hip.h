/*
 * fmd_kernels.hip.h -- gfx950 kernels of the batched FM decoder.
 *
 * Data layout in HBM (C channels, CP = C rounded up to 64):
 *   IQ input        [C][N]      complex<float>, one contiguous stream per channel (API layout)
 *   demod           [C][Mstride] complex<float>  channel-major (written coalesced by the FIR)
 *   everything else [row][CP]   "time-major": one row per sample instant, channels contiguous,
 *                               so a wavefront = 64 channels at one instant and every access is a
 *                               coalesced 256/512-byte row segment.  Buffers that feed a windowed
 *                               stage start with H "history" rows (the last H samples of the
 *                               previous call), so a window never needs a branch.
 * All positions (decimator phase, resampler fraction, tuner index, FIR ring index) are the same
 * for every channel of a batch and are tracked on the host; only signal state is per channel.
 *
 * Arithmetic is float with the reference's promotions, sequential accumulation in the
 * reference's order, and no FMA contraction (the file is compiled with -ffp-contract=off), so
 * the outputs are bit-comparable with the CPU path.  Citations: /root/reference/src/.
 */
#pragma once

// one file per stage (round 5: this was a single 4300-line header)
#include "fmd_k_common.hip.h"
#include "fmd_k_if.hip.h"
#include "fmd_k_serial.hip.h"
#include "fmd_k_rds.hip.h"
#include "fmd_k_resample.hip.h"
#include "fmd_k_tail.hip.h"
