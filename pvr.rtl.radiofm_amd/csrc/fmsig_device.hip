/*
 * fmsig_device.hip -- on-device synthetic FM IQ generator (bench / test infrastructure,
 * separate library libfmsig_hip.so; not part of the decoder).  Evaluates exactly the
 * per-sample formulas of tools/fmsig_core.h, one thread per IQ sample, so channel c of a
 * batch is the stream the host generator produces for the same parameters (SURVEY.md 8(d),
 * config 4: "device generator must equal the host generator on a spot-checked subset").
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../tools/fmsig_core.h"

__global__ __launch_bounds__(256) void k_fmsig(const fmsig_chan* __restrict__ chans,
                                               const uint8_t* __restrict__ dbits, unsigned period,
                                               uint64_t start, unsigned n, float2* __restrict__ out,
                                               size_t chan_stride)
{
  const unsigned c = blockIdx.y;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const fmsig_chan ch = chans[c];
  uint8_t a, b;
  fmsig_sample_u8(&ch, start + i, dbits + (size_t)c * period, period, &a, &b);
  out[(size_t)c * chan_stride + i] = make_float2(fmsig_u8_to_float(a), fmsig_u8_to_float(b));
}

/* the same stream as raw RTL-SDR bytes (I, Q pairs) */
__global__ __launch_bounds__(256) void k_fmsig_u8(const fmsig_chan* __restrict__ chans,
                                                  const uint8_t* __restrict__ dbits, unsigned period,
                                                  uint64_t start, unsigned n, uchar2* __restrict__ out,
                                                  size_t chan_stride)
{
  const unsigned c = blockIdx.y;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const fmsig_chan ch = chans[c];
  uint8_t a, b;
  fmsig_sample_u8(&ch, start + i, dbits + (size_t)c * period, period, &a, &b);
  out[(size_t)c * chan_stride + i] = make_uchar2(a, b);
}

extern "C" {

/* chans / dbits are DEVICE pointers (C entries, C*period bytes); out is [C][chan_stride] complex */
int fmsig_device_generate(const void* d_chans, const void* d_dbits, unsigned period, unsigned C,
                          uint64_t start, unsigned n, void* d_out, size_t chan_stride, void* stream)
{
  if (C == 0 || n == 0)
    return 0;
  if (C > 65535)
    return -1;
  hipLaunchKernelGGL(k_fmsig, dim3((n + 255) / 256, C), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const fmsig_chan*>(d_chans), static_cast<const uint8_t*>(d_dbits),
                     period, start, n, static_cast<float2*>(d_out), chan_stride);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

/* out is [C][chan_stride] byte pairs */
int fmsig_device_generate_u8(const void* d_chans, const void* d_dbits, unsigned period, unsigned C,
                             uint64_t start, unsigned n, void* d_out, size_t chan_stride, void* stream)
{
  if (C == 0 || n == 0)
    return 0;
  if (C > 65535)
    return -1;
  hipLaunchKernelGGL(k_fmsig_u8, dim3((n + 255) / 256, C), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const fmsig_chan*>(d_chans),
                     static_cast<const uint8_t*>(d_dbits), period, start, n, static_cast<uchar2*>(d_out),
                     chan_stride);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

unsigned fmsig_chan_size(void)
{
  return (unsigned)sizeof(fmsig_chan);
}
}
