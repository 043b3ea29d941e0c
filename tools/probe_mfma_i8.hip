// probe: operand layout of v_mfma_i32_16x16x64_i8 on gfx950, with exact asymmetric integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A, const int8_t* B, int* D) {  // A[16][64] row-major (m,k), B[64][16] (k,n)
    int l = threadIdx.x, r = l & 15, g = l >> 4;
    v4i a, b, c = {0, 0, 0, 0};
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; j++) { ab[j] = A[r * 64 + 16 * g + j]; bb[j] = B[(16 * g + j) * 16 + r]; }
    a = *(v4i*)ab; b = *(v4i*)bb;
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(4 * g + i) * 16 + r] = c[i];  // row = 4*(l>>4)+i, col = l&15
}
int main() {
    int8_t hA[16 * 64], hB[64 * 16]; int hD[256], ref[256];
    srand(3);
    for (auto& v : hA) v = (int8_t)(rand() % 256 - 128);
    for (auto& v : hB) v = (int8_t)(rand() % 256 - 128);
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { int s = 0; for (int kk = 0; kk < 64; kk++) s += hA[m * 64 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("mfma_i32_16x16x64_i8 natural layout: %d mismatches of 256\n", bad);
    return bad != 0;
}
