/* Exhaustive CPU check of the tile kernel's depth-key encoding (realtime_urdf_filter_amd/csrc/rtuf_kernels.hip, KeyFmt /
 * near_z_from_key): for every float window z whose 24-bit depth z24 lies in [2^(26-shift), 2^23] and every key shift the
 * library can choose (3 .. 16), the float must be recovered exactly from z24 and the float's low `shift` bits.
 * The two functions below restate the device code operation by operation (float multiply, round to nearest even).
 * Build: gcc -O2 -ffp-contract=off -o near_key_check near_key_check.c -lm ;  prints "ok <floats checked>" or the first failure. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static uint32_t z24_of(float z)                       /* rtuf_kernels.hip: z24_of */
{
  const float zc = fminf(fmaxf(z, 0.0f), 1.0f);
  return (uint32_t)lrintf(zc * 16777215.0f);
}

static float near_z_from_key(uint32_t z24, uint32_t low, int shift)      /* rtuf_kernels.hip: near_z_from_key */
{
  const uint32_t cb = f2u((float)z24 * 5.9604648328104515e-08f);
  const uint32_t span = 1u << shift;
  uint32_t cand = (cb & ~(span - 1u)) | low;
  const int d = (int)(cand - cb);
  const int half = (int)(span >> 1);
  cand = d > half ? cand - span : (d < -half ? cand + span : cand);
  return u2f(cand);
}

int main(int argc, char **argv)
{
  const int s_lo = argc > 1 ? atoi(argv[1]) : 3, s_hi = argc > 2 ? atoi(argv[2]) : 16;
  unsigned long long checked = 0;
  /* every float from the smallest z24 any shift covers up to the first float whose z24 exceeds 2^23 */
  const uint32_t zmin24 = 1u << (26 - s_hi);
  uint32_t first = f2u(((float)zmin24 - 1.0f) / 16777215.0f);
  for (uint32_t u = first; ; u++) {
    const float z = u2f(u);
    const uint32_t z24 = z24_of(z);
    if (z24 > (1u << 23)) break;
    for (int s = s_lo; s <= s_hi; s++) {
      if (z24 < (1u << (26 - s))) continue;
      const float back = near_z_from_key(z24, u & ((1u << s) - 1u), s);
      if (f2u(back) != u) { printf("FAIL shift %d z %.9g (0x%08x) z24 %u -> 0x%08x\n", s, z, u, z24, f2u(back)); return 1; }
      checked++;
    }
  }
  printf("ok %llu\n", checked);
  return 0;
}
