/*
 * fmsig.c -- host side of the synthetic FM generator (test / bench infrastructure).
 * See fmsig.h.  Build: gcc -O2 -shared -fPIC tools/fmsig.c -lm -o tools/libfmsig.so
 */
#include "fmsig.h"

#include "fmsig_core.h"

#include <string.h>

void fmsig_default(fmsig_params* p, double fs)
{
  memset(p, 0, sizeof(*p));
  p->fs = fs;
  p->f_offset = -0.15 * fs; /* LO = station + 0.15*fs */
  p->dev = 75000.0;
  p->amp = 0.5;
  p->a_mono = 0.35;
  p->a_stereo = 0.35;
  p->a_pilot = 0.09;
  p->a_rds = 0.06;
  p->f_left = 1000.0;
  p->f_right = 2500.0;
  p->noise_sigma = 0.0;
  p->seed = 7;
  p->pi = 0xD314;
  p->pty = 10;
  p->ms = 1;
  memcpy(p->ps, "TESTFM01", 8);
}

/* RDS (26,16) checkword: remainder of d(x)*x^10 by g(x)=x^10+x^8+x^7+x^5+x^4+x^3+1, plus offset */
static uint32_t rds_checkword(uint16_t data, uint32_t offset)
{
  uint32_t reg = 0;
  for (int i = 15; i >= 0; i--)
  {
    uint32_t bit = (data >> i) & 1u;
    uint32_t msb = (reg >> 9) & 1u;
    reg = (reg << 1) & 0x3FF;
    if (bit ^ msb)
      reg ^= 0x1B9;
  }
  return (reg ^ offset) & 0x3FF;
}

void fmsig_rds_groups(const fmsig_params* p, uint16_t blocks[4][4])
{
  for (int seg = 0; seg < 4; seg++)
  {
    blocks[seg][0] = p->pi;
    /* group 0A: type 0, version A, TP 0, PTY, TA 0, MS, DI bit 0, segment address */
    blocks[seg][1] = (uint16_t)(((p->pty & 0x1F) << 5) | ((p->ms & 1) << 3) | seg);
    blocks[seg][2] = 0xE0CD; /* AF: "no AF" filler pair */
    blocks[seg][3] = (uint16_t)(((uint8_t)p->ps[2 * seg] << 8) | (uint8_t)p->ps[2 * seg + 1]);
  }
}

void fmsig_rds_dbits(const fmsig_params* p, uint8_t* dbits)
{
  static const uint32_t offs[4] = {0x0FC, 0x198, 0x168, 0x1B4}; /* A, B, C, D */
  uint16_t blocks[4][4];
  uint8_t raw[416];
  fmsig_rds_groups(p, blocks);
  int k = 0;
  for (int g = 0; g < 4; g++)
    for (int b = 0; b < 4; b++)
    {
      uint32_t word = ((uint32_t)blocks[g][b] << 10) | rds_checkword(blocks[g][b], offs[b]);
      for (int i = 25; i >= 0; i--)
        raw[k++] = (word >> i) & 1u;
    }
  uint8_t d = 0;
  for (int i = 0; i < FMSIG_RDS_PERIOD_BITS; i++)
  {
    d ^= raw[i % 416];
    dbits[i] = d;
  }
}

unsigned fmsig_sched_dbits(const uint16_t* groups, unsigned ngroups, uint8_t* dbits)
{
  /* offset words A, B, C, D and C' (block 3 of version-B groups: bit 11 of block 2 set) */
  static const uint32_t offs[5] = {0x0FC, 0x198, 0x168, 0x1B4, 0x350};
  const unsigned nraw = ngroups * 104u;
  unsigned k = 0;
  uint8_t d = 0;
  /* two passes over the schedule: a period that leaves the differential encoder where it started */
  for (unsigned pass = 0; pass < 2; pass++)
    for (unsigned g = 0; g < ngroups; g++)
    {
      const uint16_t* bl = groups + 4u * g;
      const int ver_b = (bl[1] >> 11) & 1;
      for (int b = 0; b < 4; b++)
      {
        const uint32_t off = (b == 2 && ver_b) ? offs[4] : offs[b];
        const uint32_t word = ((uint32_t)bl[b] << 10) | rds_checkword(bl[b], off);
        for (int i = 25; i >= 0; i--)
        {
          d ^= (uint8_t)((word >> i) & 1u);
          dbits[k++] = d;
        }
      }
    }
  return 2u * nraw;
}

static void to_chan(const fmsig_params* p, fmsig_chan* c)
{
  c->inv_fs = 1.0 / p->fs;
  c->f_offset = p->f_offset;
  c->dev = p->dev;
  c->amp = p->amp;
  c->a_mono = p->a_mono;
  c->a_stereo = p->a_stereo;
  c->a_pilot = p->a_pilot;
  c->a_rds = p->a_rds;
  c->f_left = p->f_left;
  c->f_right = p->f_right;
  c->noise_sigma = p->noise_sigma;
  c->seed = p->seed;
}

void fmsig_generate_u8(const fmsig_params* p, uint64_t start, uint32_t n, uint8_t* iq)
{
  fmsig_chan c;
  uint8_t dbits[FMSIG_RDS_PERIOD_BITS];
  to_chan(p, &c);
  fmsig_rds_dbits(p, dbits);
  for (uint32_t i = 0; i < n; i++)
    fmsig_sample_u8(&c, start + i, dbits, FMSIG_RDS_PERIOD_BITS, &iq[2 * i], &iq[2 * i + 1]);
}

void fmsig_u8_to_f32(const uint8_t* iq, uint32_t nbytes, float* out)
{
  for (uint32_t i = 0; i < nbytes; i++)
    out[i] = fmsig_u8_to_float(iq[i]);
}

void fmsig_generate_f32(const fmsig_params* p, uint64_t start, uint32_t n, float* out)
{
  fmsig_chan c;
  uint8_t dbits[FMSIG_RDS_PERIOD_BITS];
  to_chan(p, &c);
  fmsig_rds_dbits(p, dbits);
  for (uint32_t i = 0; i < n; i++)
  {
    uint8_t a, b;
    fmsig_sample_u8(&c, start + i, dbits, FMSIG_RDS_PERIOD_BITS, &a, &b);
    out[2 * i] = fmsig_u8_to_float(a);
    out[2 * i + 1] = fmsig_u8_to_float(b);
  }
}

void fmsig_generate_f32_bits(const fmsig_params* p, const uint8_t* dbits, unsigned period_bits, uint64_t start,
                             uint32_t n, float* out)
{
  fmsig_chan c;
  to_chan(p, &c);
  for (uint32_t i = 0; i < n; i++)
  {
    uint8_t a, b;
    fmsig_sample_u8(&c, start + i, dbits, period_bits, &a, &b);
    out[2 * i] = fmsig_u8_to_float(a);
    out[2 * i + 1] = fmsig_u8_to_float(b);
  }
}
