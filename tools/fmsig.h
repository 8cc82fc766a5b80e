/*
 * fmsig.h -- deterministic synthetic broadcast-FM IQ generator (test / bench infrastructure).
 *
 * Not part of the product path and not derived from the reference (the reference has no
 * signal generator; SURVEY.md section 8(c) describes the recipe this follows).
 *
 * Signal:  m(t) = a_mono*(L+R)/2 + a_stereo*(L-R)/2*sin(2pi*38k*t) + a_pilot*sin(2pi*19k*t)
 *                 + a_rds*rds(t)*sin(2pi*57k*t)
 *          L = sin(2pi*f_left*t), R = sin(2pi*f_right*t)
 *          rds(t) = +-sin(2pi*u), u = phase inside the 1/1187.5 s bit, sign = differentially
 *          encoded bit of a repeating sequence of 0A groups (PI, PTY, MS, 8-char PS).
 *          IQ = amp*exp(j*2pi*(f_offset*t + dev*Int m)) + N(0, sigma) per rail, quantised to
 *          u8 like an RTL-SDR and mapped back with the reference's formula
 *          (RTL_SDR_Source.cpp:209-210).
 * The FM phase integral is closed-form in t, so any sample can be generated independently
 * (the HIP generator in pvr.rtl.radiofm_amd/csrc/fmsig_device.hip evaluates the same formulas).
 */
#ifndef FMSIG_H
#define FMSIG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FMSIG_RDS_PERIOD_BITS 832 /* 2 x (4 groups x 104 bits): covers odd overall parity */

typedef struct fmsig_params
{
  double fs;
  double f_offset;
  double dev;
  double amp;
  double a_mono, a_stereo, a_pilot, a_rds;
  double f_left, f_right;
  double noise_sigma;
  uint64_t seed;
  uint16_t pi;
  uint8_t pty;
  uint8_t ms;
  char ps[8];
} fmsig_params;

/* fills the reference-style defaults for a stereo + RDS station at IF rate fs */
void fmsig_default(fmsig_params* p, double fs);

/* differentially encoded RDS bit table (one byte per bit, 0/1), FMSIG_RDS_PERIOD_BITS long */
void fmsig_rds_dbits(const fmsig_params* p, uint8_t* dbits);
/* the undifferentiated 104-bit groups (for tests): 4 groups x 4 blocks x 26 bits */
void fmsig_rds_groups(const fmsig_params* p, uint16_t blocks[4][4]);

/* A station with a group schedule of its own: `ngroups` groups of four 16-bit blocks, sent in a loop.
 * Version-B groups (bit 11 of block 2) get offset word C' on block 3.  Writes the differentially encoded
 * bit table of TWO passes (2 * ngroups * 104 entries: the encoder is back at its start after an even number
 * of passes whatever the schedule's parity) and returns its length. */
unsigned fmsig_sched_dbits(const uint16_t* groups, unsigned ngroups, uint8_t* dbits);
/* like fmsig_generate_f32 with the station's RDS bits taken from a table of the caller's */
void fmsig_generate_f32_bits(const fmsig_params* p, const uint8_t* dbits, unsigned period_bits, uint64_t start,
                             uint32_t n, float* iq_f32);

/* generate n IQ samples starting at absolute sample index start */
void fmsig_generate_u8(const fmsig_params* p, uint64_t start, uint32_t n, uint8_t* iq_u8);
void fmsig_generate_f32(const fmsig_params* p, uint64_t start, uint32_t n, float* iq_f32);
/* u8 -> float with the reference's conversion */
void fmsig_u8_to_f32(const uint8_t* iq_u8, uint32_t n_bytes, float* out);

#ifdef __cplusplus
}
#endif
#endif
