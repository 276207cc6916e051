/* CPU check of the bit-pattern identities the round-6 tile kernel uses in place of 4-cycle-class instructions
 * (realtime_urdf_filter_amd/csrc/rtuf_kernels.hip, RTUF_FAST_CLASS; DESIGN.md section 4, docs/experiments.md R6.7).
 * Every function here restates a device function; tests/test_fast_class_cpu.py checks that the constants are the device code's.
 *   1. z24_of_upper_half(z) == z24_of(z)                       for EVERY float z > 0.5 (incl. > 1, +inf); the kernels use it for z >= 0.51
 *   2. bits(z24 + 0x3E800001) == (float)(z24 + 1) * 2^-24      for every 24-bit depth >= 2^23 - 1
 *   3. the 32-bit edge constant == the low 32 bits of the 64-bit one, and the shift form of the inclusive-edge bias == its
 *      comparison form                                         for 120 M vertex pairs (random, near-coincident, axis-aligned)
 *   4. the walk's "right-hand pixel is outside the box" == wrap && odd width    for every box width 1..64
 * gcc -O2 -ffp-contract=off; prints "ok <cases>" or the first counter-example. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float flt(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* device: z24_of */
static uint32_t z24_of(float z)
{
  const float zc = fminf(fmaxf(z, 0.0f), 1.0f);
  return (uint32_t)(int32_t)rintf(zc * 16777215.0f);
}
/* device: z24_of_upper_half */
static uint32_t z24_of_upper_half(float z)
{
  const float zc = fminf(fmaxf(z, 0.0f), 1.0f);
  return bits(zc * 16777215.0f) - 0x4A800000u;
}
static int mul24(int a, int b)
{
  const int64_t x = ((int64_t)((int32_t)((uint32_t)a << 8)) >> 8), y = ((int64_t)((int32_t)((uint32_t)b << 8)) >> 8);
  return (int)(uint32_t)(uint64_t)(x * y);
}

int main(void)
{
  unsigned long long cases = 0;
  /* 1 */
  for (uint32_t u = 0x3F000001u; u <= 0x7F800000u; u++) {      /* (0.5 itself: p = 8388607.5 is below 2^23, a tie the pattern does not round) */
    const float z = flt(u);
    if (z24_of(z) != z24_of_upper_half(z)) { printf("z24: z bits %08x: %u vs %u\n", u, z24_of(z), z24_of_upper_half(z)); return 1; }
    cases++;
  }
  /* 2 */
  for (uint32_t k = 8388607u; k <= 16777215u; k++) {
    const float a = (float)(k + 1u) * 5.9604644775390625e-08f;
    if (bits(a) != k + 0x3E800001u) { printf("z from key: %u\n", k); return 1; }
    cases++;
  }
  /* 3 */
  uint64_t s = 88172645463325252ull;
  for (long it = 0; it < 120000000L; it++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint64_t t = s * 0x9E3779B97F4A7C15ull;
    const int mode = (int)((s >> 60) & 7);
    int xi = (int)(s & 0xfffff) - 1024, yi = (int)((s >> 20) & 0xfffff) - 1024, xj, yj;
    if (mode == 0) { xj = (int)(t & 0xfffff) - 1024; yj = (int)((t >> 20) & 0xfffff) - 1024; }
    else {
      const int r = mode == 1 ? 2047 : (mode == 2 ? 255 : (mode == 3 ? 31 : 3));
      xj = xi + (mode == 5 ? 0 : (int)(t & (uint64_t)(2 * r + 1)) - r); yj = yi + (mode == 6 ? 0 : (int)((t >> 24) & (uint64_t)(2 * r + 1)) - r);
      if (xj < -1024) xj = -1024; if (yj < -1024) yj = -1024; if (xj > 1047551) xj = 1047551; if (yj > 1047551) yj = 1047551;
    }
    const int dcdx = yi - yj, dcdy = xi - xj;
    long long c = (long long)dcdx * xi - (long long)dcdy * yi;
    const int bias_cmp = (dcdx < 0 || (dcdx == 0 && dcdy > 0)) ? 1 : 0;
    c += bias_cmp;
    const int C64 = (int)(-((-c) >> 8));
    const int X = xi >> 8, Y = yi >> 8, xf = xi & 255, yf = yi & 255;
    const uint32_t bias = ((uint32_t)(dcdx + dcdx) - ((uint32_t)(0 - dcdy) >> 31)) >> 31;
    const int tt = mul24(dcdx, xf) - mul24(dcdy, yf) + (int)bias;
    const int C32 = (int)((uint32_t)mul24(dcdx, X) - (uint32_t)mul24(dcdy, Y) + (uint32_t)(-((-tt) >> 8)));
    if ((int)bias != bias_cmp || C64 != C32) { printf("edge: %d %d %d %d: %d vs %d, bias %u vs %d\n", xi, yi, xj, yj, C64, C32, bias, bias_cmp); return 1; }
    cases++;
  }
  /* 4 */
  for (int w = 1; w <= 64; w++) {
    const int lx0 = 0, lx1 = w - 1, qcols = (lx1 - lx0 + 2) >> 1, back = 2 * (qcols - 1), odd_w = ((lx1 - lx0) & 1) == 0;
    for (int px = lx0; px <= lx0 + back; px += 2) {
      const int right_old = px < lx1, wrap = px == lx0 + back;
      if (right_old != !(wrap && odd_w)) { printf("right: w %d px %d\n", w, px); return 1; }
      cases++;
    }
  }
  printf("ok %llu\n", cases);
  return 0;
}
