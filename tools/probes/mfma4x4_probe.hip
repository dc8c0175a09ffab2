// Probe: operand/broadcast semantics of v_mfma_f32_4x4x1_16b_f32 with CBSZ/ABID/BLGP on gfx950.
// Model under test (lane = 4*block + i, block = 8x + 4y + z):
//   A_block(x,y,z)[i] = a[lane(x, y, abid, i)]          (cbsz = 2: groups of 4 consecutive blocks share block `abid`;
//                                                         cbsz = 3: groups of 8, abid = 0..7; cbsz = 4: all 16, abid = 0..15)
//   B_block(x,y,z)[j] = b[lane(blgp==1 ? 0 : 1, y, z, j)] (blgp = 1: lanes 0-31 -> 32-63; blgp = 2: lanes 32-63 -> 0-31)
//   D[reg i][lane(block, j)] += A_block[i] * B_block[j]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID, int BLGP> __global__ void k(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, BLGP);
    for (int r = 0; r < 4; r++) d[r * 64 + l] = acc[r];
}
template <int CBSZ, int ABID, int BLGP> int run(const float *da, const float *db, float *dd, const float *ha, const float *hb) {
    hipLaunchKernelGGL((k<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, da, db, dd);
    float hd[256];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; blk++)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                int ablk = blk, bblk = blk;
                if (CBSZ == 2) ablk = (blk & ~3) | ABID;
                if (CBSZ == 3) ablk = (blk & ~7) | ABID;
                if (CBSZ == 4) ablk = ABID; // all 16 blocks read block `abid`
                if (BLGP == 1) bblk = blk & 7;
                if (BLGP == 2) bblk = (blk & 7) | 8;
                const float want = ha[4 * ablk + i] * hb[4 * bblk + j];
                const float got = hd[i * 64 + 4 * blk + j];
                if (want != got) bad++;
            }
    printf("cbsz=%d abid=%d blgp=%d: %s (%d mismatches)\n", CBSZ, ABID, BLGP, bad ? "MODEL WRONG" : "model ok", bad);
    return bad;
}
int main() {
    float ha[64], hb[64];
    for (int i = 0; i < 64; i++) {
        ha[i] = (float)(1 + i);
        hb[i] = (float)(101 + 3 * i);
    }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<0, 0, 0>(da, db, dd, ha, hb);
    bad += run<2, 0, 0>(da, db, dd, ha, hb);
    bad += run<2, 3, 0>(da, db, dd, ha, hb);
    bad += run<0, 0, 1>(da, db, dd, ha, hb);
    bad += run<0, 0, 2>(da, db, dd, ha, hb);
    bad += run<2, 1, 1>(da, db, dd, ha, hb);
    bad += run<2, 2, 2>(da, db, dd, ha, hb);
    bad += run<3, 0, 0>(da, db, dd, ha, hb);
    bad += run<3, 5, 1>(da, db, dd, ha, hb);
    bad += run<3, 7, 2>(da, db, dd, ha, hb);
    bad += run<4, 0, 0>(da, db, dd, ha, hb);
    bad += run<4, 9, 0>(da, db, dd, ha, hb);
    bad += run<4, 15, 0>(da, db, dd, ha, hb);
    return bad ? 1 : 0;
}
