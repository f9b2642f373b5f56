// Probe: issue cost of single VALU instructions on gfx950 (one wave per SIMD, eight independent chains per lane, 4096 x 8
// instructions per lane), relative to v_xor_b32.  build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o tools/probes/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN8(INS)                                                                                                   \
  for (int i = 0; i < 4096; ++i) {                                                                                    \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                              \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])      \
                 : "v"(c), "s"(k));                                                                                   \
  }
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n\t"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n\t"
#define I_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n\t"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n\t"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %" #n "\n\t"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 13\n\t"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 13, %" #n "\n\t"
#define I_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n\t"
#define I_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %" #n "\n\t"
#define I_XOR3(n) "v_add3_u32 %" #n ", %" #n ", %8, %" #n "\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n\t"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %" #n "\n\t"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %" #n ", %" #n "\n\t"
#define I_CVTPK(n) "v_cvt_pk_bf16_f32 %" #n ", %" #n ", %8\n\t"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 16\n\t"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %" #n "\n\t"
#define I_MADU64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %8, %" #n "\n\t"
template <int W> __global__ void __launch_bounds__(256) rate_k(unsigned *out, unsigned k) {
  unsigned a[8]; unsigned c = threadIdx.x * 2654435761u + 12345u;
  for (int j = 0; j < 8; ++j) a[j] = c + j * 77u;
  if (W == 0) { CHAIN8(I_XOR) } if (W == 1) { CHAIN8(I_MULLO) } if (W == 2) { CHAIN8(I_MULHI) } if (W == 3) { CHAIN8(I_MUL24) }
  if (W == 4) { CHAIN8(I_MAD24) } if (W == 5) { CHAIN8(I_ALIGN) } if (W == 6) { CHAIN8(I_LSHR) } if (W == 7) { CHAIN8(I_PKMUL) }
  if (W == 8) { CHAIN8(I_XAD) } if (W == 9) { CHAIN8(I_XOR3) } if (W == 10) { CHAIN8(I_LSHLADD) } if (W == 11) { CHAIN8(I_EXP) }
  if (W == 12) { CHAIN8(I_RCP) } if (W == 13) { CHAIN8(I_FMA) } if (W == 14) { CHAIN8(I_CVTPK) } if (W == 15) { CHAIN8(I_PERM) }
  if (W == 16) { CHAIN8(I_BFE) } if (W == 17) { CHAIN8(I_ANDOR) }
  unsigned s = 0; for (int j = 0; j < 8; ++j) s ^= a[j];
  if (s == 0x1234567u) out[0] = s;
}
template <int W> float run(unsigned *out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_k<W><<<256, 256>>>(out, 0x9E3779B1u); hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; ++r) { hipEventRecord(e0); rate_k<W><<<256, 256>>>(out, 0x9E3779B1u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
  return best * 1e3f;
}
int main() {
  unsigned *out; hipMalloc(&out, 64);
  const char *nm[18] = {"v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_alignbit_b32", "v_lshrrev_b32", "v_pk_mul_lo_u16",
                        "v_xad_u32", "v_add3_u32", "v_lshl_add_u32", "v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_cvt_pk_bf16_f32", "v_perm_b32", "v_bfe_u32", "v_and_or_b32"};
  float t[18] = {run<0>(out), run<1>(out), run<2>(out), run<3>(out), run<4>(out), run<5>(out), run<6>(out), run<7>(out), run<8>(out), run<9>(out),
                 run<10>(out), run<11>(out), run<12>(out), run<13>(out), run<14>(out), run<15>(out), run<16>(out), run<17>(out)};
  for (int i = 0; i < 18; ++i) printf("%-20s %8.1f us  = %.2f x v_xor_b32\n", nm[i], t[i], t[i] / t[0]);
  return 0;
}
