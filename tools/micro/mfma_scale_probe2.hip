// Second probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3): the first one (mfma_scale_probe.hip) found that neither "32
// consecutive k per lane" nor "4 x 8 interleaved" reproduces a host sum once the block scales vary.  This one measures, with
// one-hot operands, (1) which register position (lane group g = lane >> 4, byte j of the 8 dwords) of operand A pairs with
// which position of operand B, and (2) which lane group's scale register applies to each position - everything a kernel needs:
// absolute k never matters, only the pairing and the scale blocks.  Then it checks the derived map with random data and scales.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void mfma_once(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}
static uint8_t ha[64][32], hb[64][32];
static int hsa[64], hsb[64];
static v4f hc[64];
static v8i *da, *db; static int *dsa, *dsb; static v4f* dc;
static void run() {
  CK(hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
}
static float D(int m, int n) { return hc[(m >> 2) * 16 + n][m & 3]; }   // D[row m][col n]: lane 16 (m / 4) + n, register m % 4
static const uint8_t ONE = 0x38;   // e4m3 1.0

int main() {
  CK(hipMalloc(&da, 2048)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 1024));
  // 0. sanity: all ones, unit scales -> 128 everywhere
  memset(ha, ONE, sizeof(ha)); memset(hb, ONE, sizeof(hb));
  for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
  run();
  printf("all ones, unit scales: D[0][0] = %g, D[5][9] = %g (expect 128)\n", D(0, 0), D(5, 9));
  // 0b. row / column orientation: A row 3 all ones, B col 5 all ones
  memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
  for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; ++j) { ha[16 * g + 3][j] = ONE; hb[16 * g + 5][j] = ONE; }
  run();
  printf("A row 3 x B col 5: D[3][5] = %g, D[5][3] = %g (expect 128, 0)\n", D(3, 5), D(5, 3));
  // 1. pairing: A one-hot at (row 3, group g, byte j) against B one-hot at (col 5, group g2, byte j2)
  int pair_g[4][32], pair_j[4][32];
  for (int g = 0; g < 4; ++g)
    for (int j = 0; j < 32; ++j) {
      pair_g[g][j] = pair_j[g][j] = -1;
      int fg = -1;
      for (int g2 = 0; g2 < 4; ++g2) {   // stage 1: the B lane group (all of its bytes ones)
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        ha[16 * g + 3][j] = ONE;
        for (int jj = 0; jj < 32; ++jj) hb[16 * g2 + 5][jj] = ONE;
        run();
        if (D(3, 5) == 1.0f) { fg = g2; break; }
      }
      if (fg < 0) continue;
      for (int j2 = 0; j2 < 32; ++j2) {   // stage 2: the byte
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        ha[16 * g + 3][j] = ONE; hb[16 * fg + 5][j2] = ONE;
        run();
        if (D(3, 5) == 1.0f) { pair_g[g][j] = fg; pair_j[g][j] = j2; break; }
      }
    }
  int ident = 1;
  for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; ++j) if (pair_g[g][j] != g || pair_j[g][j] != j) ident = 0;
  printf("pairing of A positions with B positions: %s\n", ident ? "identity (same lane group, same byte)" : "NOT the identity:");
  if (!ident) for (int g = 0; g < 4; ++g) { printf("  A group %d:", g); for (int j = 0; j < 32; ++j) printf(" (%d,%d)", pair_g[g][j], pair_j[g][j]); printf("\n"); }
  // 2. which lane group's scale applies to position (g, j): one-hot on one side, all ones on the other, scales 2^(gs + 1) in lane group gs
  int sgrpA[4][32], sgrpB[4][32];
  for (int side = 0; side < 2; ++side)
    for (int g = 0; g < 4; ++g)
      for (int j = 0; j < 32; ++j) {
        memset(ha, side == 0 ? 0 : ONE, sizeof(ha)); memset(hb, side == 0 ? ONE : 0, sizeof(hb));
        for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
        if (side == 0) { ha[16 * g + 3][j] = ONE; for (int gs = 0; gs < 4; ++gs) hsa[16 * gs + 3] = 127 + gs + 1; }
        else { hb[16 * g + 5][j] = ONE; for (int gs = 0; gs < 4; ++gs) hsb[16 * gs + 5] = 127 + gs + 1; }
        run();
        const float v = D(3, 5);
        int gs = -1;
        for (int t = 0; t < 4; ++t) if (v == ldexpf(1.f, t + 1)) gs = t;
        (side == 0 ? sgrpA : sgrpB)[g][j] = gs;
        if (gs < 0 && g == 0 && j < 2) printf("  (side %d g %d j %d: D[3][5] = %g)\n", side, g, j, v);
      }
  for (int side = 0; side < 2; ++side) {
    printf("scale lane group that applies to operand %c position (g, byte j):\n", side == 0 ? 'A' : 'B');
    for (int g = 0; g < 4; ++g) { printf("  g = %d:", g); for (int j = 0; j < 32; ++j) printf(" %d", (side == 0 ? sgrpA : sgrpB)[g][j]); printf("\n"); }
  }
  // 3. random check with the derived map: memory k of block s = the 32 positions whose scale group is s, in (g, j) order
  int kofA[4][32]; int cnt[4] = {0, 0, 0, 0};
  for (int g = 0; g < 4; ++g) for (int j = 0; j < 32; ++j) { const int s = sgrpA[g][j]; kofA[g][j] = s < 0 ? 0 : 32 * s + (cnt[s]++ & 31); }
  printf("positions per scale block: %d %d %d %d\n", cnt[0], cnt[1], cnt[2], cnt[3]);
  static float A[16][128], B[128][16]; static int SA[16][4], SB[16][4];
  srand(11);
  auto enc = [](int v) -> uint8_t {   // small integers as e4m3
    const uint8_t s = v < 0 ? 0x80 : 0; int a = abs(v);
    if (a == 0) return s;
    int e = 0; while ((a >> (e + 1)) > 0) ++e;   // a in [2^e, 2^(e+1))
    const int m = ((a << 3) >> e) & 7;            // exact for a <= 15
    return s | (uint8_t)(((e + 7) << 3) | m);
  };
  for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; ++k) { A[r][k] = (float)(rand() % 17 - 8); B[k][r] = (float)(rand() % 17 - 8); }
  for (int r = 0; r < 16; ++r) for (int s = 0; s < 4; ++s) { SA[r][s] = 127 + rand() % 7 - 3; SB[r][s] = 127 + rand() % 7 - 3; }
  for (int l = 0; l < 64; ++l) {
    const int rc = l & 15, g = l >> 4;
    for (int j = 0; j < 32; ++j) {
      const int k = kofA[g][j];
      ha[l][j] = enc((int)A[rc][k]);
      const int pg = pair_g[g][j] < 0 ? g : pair_g[g][j], pj = pair_j[g][j] < 0 ? j : pair_j[g][j];
      hb[16 * pg + rc][pj] = enc((int)B[k][rc]);
    }
    hsa[l] = SA[rc][g]; hsb[l] = SB[rc][g];
  }
  run();
  int bad = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double ref = 0;
    for (int k = 0; k < 128; ++k) ref += (double)A[m][k] * ldexp(1.0, SA[m][k / 32] - 127) * B[k][n] * ldexp(1.0, SB[n][k / 32] - 127);
    if (fabs(D(m, n) - ref) > 1e-4 * (1 + fabs(ref))) { if (bad < 4) printf("  D[%d][%d] = %g, host %g\n", m, n, D(m, n), ref); ++bad; }
  }
  printf("random data + random block scales through the derived map: %d / 256 mismatches\n", bad);
  return 0;
}
