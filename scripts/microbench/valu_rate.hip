// Issue-rate microbenchmark: cycles per wave64 instruction per SIMD for the instruction kinds the traversal
// node step is made of, at 1 / 2 / 4 / 8 waves per SIMD.  Each kernel runs LOOPS x 32 independent copies of one
// instruction; time by s_memtime inside the wave (shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEFINE_KERNEL(NAME, ASM)                                                                    \
  __global__ void __launch_bounds__(1024) NAME(int loops_in, unsigned long long* out, float* sink) {    \
    __shared__ float lds_buf[8192]; int loops = loops_in; if (loops == 123456789) lds_buf[threadIdx.x] = 1.0f; float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f;                                            \
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3; \
    if (loops < -(1 << 24)) { r0 = __int_as_float((int)(threadIdx.x * 8)); loops = -loops - (1 << 24); } else if (loops < 0) { r0 = __int_as_float((int)(threadIdx.x * 4)); loops = -loops; }                                               \
    unsigned long long t0 = __builtin_readcyclecounter();                                           \
    for (int i = 0; i < loops; ++i) {                                                               \
      asm volatile(REP8(ASM) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"); \
    }                                                                                               \
    unsigned long long t1 = __builtin_readcyclecounter();                                           \
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;       \
    if (r0 + r1 + r2 + r3 == 12345.678f) sink[0] = r0;                                              \
  }

// each ASM string = 4 independent instructions (so REP8 -> 32 per loop iteration)
DEFINE_KERNEL(k_fma,      "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
DEFINE_KERNEL(k_mul,      "v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_mul_f32_e32 %3, %4, %3\n")
DEFINE_KERNEL(k_cvtub,    "v_cvt_f32_ubyte1_e32 v100, %0\n v_cvt_f32_ubyte1_e32 v101, %1\n v_cvt_f32_ubyte1_e32 v102, %2\n v_cvt_f32_ubyte1_e32 v103, %3\n")
DEFINE_KERNEL(k_max3,     "v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n")
DEFINE_KERNEL(k_min,      "v_min_f32_e32 %0, %4, %0\n v_min_f32_e32 %1, %4, %1\n v_min_f32_e32 %2, %4, %2\n v_min_f32_e32 %3, %4, %3\n")
DEFINE_KERNEL(k_cnd64,    "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]\n")
DEFINE_KERNEL(k_cnd32,    "v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n")
DEFINE_KERNEL(k_cmp64,    "v_cmp_le_f32_e64 s[20:21], %0, %4\n v_cmp_le_f32_e64 s[22:23], %1, %4\n v_cmp_le_f32_e64 s[20:21], %2, %4\n v_cmp_le_f32_e64 s[22:23], %3, %4\n")
DEFINE_KERNEL(k_cmp32,    "v_cmp_lt_u32_e32 vcc, %0, %4\n v_cmp_lt_u32_e32 vcc, %1, %4\n v_cmp_lt_u32_e32 vcc, %2, %4\n v_cmp_lt_u32_e32 vcc, %3, %4\n")
DEFINE_KERNEL(k_and,      "v_and_b32_e32 %0, %4, %0\n v_and_b32_e32 %1, %4, %1\n v_and_b32_e32 %2, %4, %2\n v_and_b32_e32 %3, %4, %3\n")
DEFINE_KERNEL(k_lshlor,   "v_lshl_or_b32 %0, %0, 2, %4\n v_lshl_or_b32 %1, %1, 2, %4\n v_lshl_or_b32 %2, %2, 2, %4\n v_lshl_or_b32 %3, %3, 2, %4\n")
DEFINE_KERNEL(k_minu,     "v_min_u32_e32 %0, %4, %0\n v_min_u32_e32 %1, %4, %1\n v_min_u32_e32 %2, %4, %2\n v_min_u32_e32 %3, %4, %3\n")
DEFINE_KERNEL(k_mov,      "v_mov_b32_e32 v100, %0\n v_mov_b32_e32 v101, %1\n v_mov_b32_e32 v102, %2\n v_mov_b32_e32 v103, %3\n")
DEFINE_KERNEL(k_pkfma,    "v_pk_fma_f32 v[100:101], v[100:101], v[104:105], v[106:107]\n v_pk_fma_f32 v[102:103], v[102:103], v[104:105], v[106:107]\n v_pk_fma_f32 v[100:101], v[100:101], v[104:105], v[106:107]\n v_pk_fma_f32 v[102:103], v[102:103], v[104:105], v[106:107]\n")
DEFINE_KERNEL(k_pkmul,    "v_pk_mul_f32 v[100:101], v[100:101], v[104:105]\n v_pk_mul_f32 v[102:103], v[102:103], v[104:105]\n v_pk_mul_f32 v[100:101], v[100:101], v[104:105]\n v_pk_mul_f32 v[102:103], v[102:103], v[104:105]\n")
DEFINE_KERNEL(k_sand,     "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[20:21], s[22:23]\n")
DEFINE_KERNEL(k_mix,      "v_fma_f32 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cvt_f32_ubyte1_e32 v100, %2\n v_cmp_le_f32_e64 s[22:23], %3, %4\n")


#define K4(NAME, OP) DEFINE_KERNEL(NAME, OP " %0, %4, %0\n " OP " %1, %4, %1\n " OP " %2, %4, %2\n " OP " %3, %4, %3\n")
#define K4_3(NAME, OP) DEFINE_KERNEL(NAME, OP " %0, %0, %4, %5\n " OP " %1, %1, %4, %5\n " OP " %2, %2, %4, %5\n " OP " %3, %3, %4, %5\n")
#define K4_1(NAME, OP) DEFINE_KERNEL(NAME, OP " v100, %0\n " OP " v101, %1\n " OP " v102, %2\n " OP " v103, %3\n")
K4(k_addu, "v_add_u32_e32")
K4(k_subf, "v_sub_f32_e32")
K4(k_addf, "v_add_f32_e32")
K4(k_maxf, "v_max_f32_e32")
K4(k_or, "v_or_b32_e32")
K4(k_xor, "v_xor_b32_e32")
K4(k_shl, "v_lshlrev_b32_e32")
K4(k_shr, "v_lshrrev_b32_e32")
K4(k_ashr, "v_ashrrev_i32_e32")
K4(k_mul24, "v_mul_u32_u24_e32")
K4(k_maxu, "v_max_u32_e32")
K4(k_mini, "v_min_i32_e32")
K4_3(k_bfe, "v_bfe_u32")
K4_3(k_perm, "v_perm_b32")
K4_3(k_andor, "v_and_or_b32")
K4_3(k_add3, "v_add3_u32")
K4_3(k_lshladd, "v_lshl_add_u32")
K4_3(k_mad24, "v_mad_u32_u24")
K4_3(k_med3, "v_med3_f32")
K4_3(k_min3u, "v_min3_u32")
K4_3(k_fma_b, "v_fma_f32")
K4_3(k_alignbit, "v_alignbit_b32")
K4_1(k_cvtu32, "v_cvt_f32_u32_e32")
K4_1(k_cvtub0, "v_cvt_f32_ubyte0_e32")
K4_1(k_rcp, "v_rcp_f32_e32")
K4_1(k_cvtf16, "v_cvt_f32_f16_e32")
DEFINE_KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa v100, %0 src0_sel:BYTE_1\n v_cvt_f32_u32_sdwa v101, %1 src0_sel:BYTE_2\n v_cvt_f32_u32_sdwa v102, %2 src0_sel:BYTE_1\n v_cvt_f32_u32_sdwa v103, %3 src0_sel:BYTE_3\n")
DEFINE_KERNEL(k_mul_sdwa, "v_mul_f32_sdwa v100, %0, %4 src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa v101, %1, %4 src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa v102, %2, %4 src0_sel:DWORD src1_sel:DWORD\n v_mul_f32_sdwa v103, %3, %4 src0_sel:DWORD src1_sel:DWORD\n")
DEFINE_KERNEL(k_fmac, "v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5\n")
DEFINE_KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x3f800008, %5\n v_fmamk_f32 %1, %1, 0x3f800008, %5\n v_fmamk_f32 %2, %2, 0x3f800008, %5\n v_fmamk_f32 %3, %3, 0x3f800008, %5\n")
DEFINE_KERNEL(k_cmp_f_e32, "v_cmp_le_f32_e32 vcc, %0, %4\n v_cmp_le_f32_e32 vcc, %1, %4\n v_cmp_le_f32_e32 vcc, %2, %4\n v_cmp_le_f32_e32 vcc, %3, %4\n")
DEFINE_KERNEL(k_cnd_vccw, "v_cmp_lt_u32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cmp_lt_u32_e32 vcc, %2, %4\n v_cndmask_b32_e32 %3, %3, %4, vcc\n")
DEFINE_KERNEL(k_dswrite, "ds_write_b32 %0, %4\n ds_write_b32 %0, %5 offset:1024\n ds_write_b32 %0, %4 offset:2048\n ds_write_b32 %0, %5 offset:3072\n")
DEFINE_KERNEL(k_dsread, "ds_read_b32 v100, %0\n ds_read_b32 v101, %0 offset:1024\n ds_read_b32 v102, %0 offset:2048\n ds_read_b32 v103, %0 offset:3072\n s_waitcnt lgkmcnt(0)\n")

// r03 additions: candidates for an 8-wide / octant-ordered node step
DEFINE_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %4, %5 bitop3:0x20\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x20\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x20\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x20\n")
K4(k_bcnt, "v_bcnt_u32_b32")
K4_1(k_ffbl, "v_ffbl_b32_e32")
K4(k_bfm, "v_bfm_b32")
K4_3(k_addlshl, "v_add_lshl_u32")
K4_3(k_fmamix, "v_fma_mix_f32")
DEFINE_KERNEL(k_fmamix_hi, "v_fma_mix_f32 %0, %0, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n")
DEFINE_KERNEL(k_cvtpkfp8, "v_cvt_pk_f32_fp8 v[100:101], %0\n v_cvt_pk_f32_fp8 v[102:103], %1\n v_cvt_pk_f32_fp8 v[104:105], %2\n v_cvt_pk_f32_fp8 v[106:107], %3\n")
K4_3(k_pkfmaf16, "v_pk_fma_f16")
K4(k_pkmaxf16, "v_pk_max_f16")
K4(k_pkminf16, "v_pk_min_f16")
K4(k_mullo, "v_mul_lo_u32")
K4_3(k_dot2, "v_dot2_f32_f16")
K4_1(k_not, "v_not_b32_e32")
DEFINE_KERNEL(k_dsread_u8, "ds_read_u8 v100, %0\n ds_read_u8 v101, %0 offset:1024\n ds_read_u8 v102, %0 offset:2048\n ds_read_u8 v103, %0 offset:3072\n s_waitcnt lgkmcnt(0)\n")
DEFINE_KERNEL(k_dsread_u16, "ds_read_u16 v100, %0\n ds_read_u16 v101, %0 offset:1024\n ds_read_u16 v102, %0 offset:2048\n ds_read_u16 v103, %0 offset:3072\n s_waitcnt lgkmcnt(0)\n")
DEFINE_KERNEL(k_dswrite_b64, "ds_write_b64 %0, v[100:101]\n ds_write_b64 %0, v[102:103] offset:2048\n ds_write_b64 %0, v[100:101] offset:4096\n ds_write_b64 %0, v[102:103] offset:6144\n")
DEFINE_KERNEL(k_dsread_b64, "ds_read_b64 v[100:101], %0\n ds_read_b64 v[102:103], %0 offset:2048\n ds_read_b64 v[104:105], %0 offset:4096\n ds_read_b64 v[106:107], %0 offset:6144\n s_waitcnt lgkmcnt(0)\n")

// r05 additions: the rest of the render kernels' opcodes (IEEE division / sqrt expansions of k_shade, the any-hit set-up),
// so that scripts/valu_class_calib.sh can say which hardware class counter each of them ticks and at what rate it issues
DEFINE_KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %4, %5\n v_div_scale_f32 %1, vcc, %1, %4, %5\n v_div_scale_f32 %2, vcc, %2, %4, %5\n v_div_scale_f32 %3, vcc, %3, %4, %5\n")
K4_3(k_divfmas, "v_div_fmas_f32")
K4_3(k_divfixup, "v_div_fixup_f32")
K4_1(k_sqrt, "v_sqrt_f32_e32")
K4_1(k_rsq, "v_rsq_f32_e32")
DEFINE_KERNEL(k_cmpclass, "v_cmp_class_f32_e32 vcc, %0, %4\n v_cmp_class_f32_e32 vcc, %1, %4\n v_cmp_class_f32_e32 vcc, %2, %4\n v_cmp_class_f32_e32 vcc, %3, %4\n")
K4_1(k_rndne, "v_rndne_f32_e32")
K4_1(k_cvti32, "v_cvt_i32_f32_e32")
K4_1(k_cvtu32f, "v_cvt_u32_f32_e32")
DEFINE_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_lo_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_lo_u32_b32 %3, %4, %3\n")
DEFINE_KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4\n")
K4_3(k_min3f, "v_min3_f32")
DEFINE_KERNEL(k_cvtscalefp8, "v_cvt_scalef32_pk_f32_fp8 v[100:101], %0, %4\n v_cvt_scalef32_pk_f32_fp8 v[102:103], %1, %4\n v_cvt_scalef32_pk_f32_fp8 v[104:105], %2, %4\n v_cvt_scalef32_pk_f32_fp8 v[106:107], %3, %4\n")
DEFINE_KERNEL(k_lshladd64, "v_lshl_add_u64 v[100:101], v[102:103], 2, v[104:105]\n v_lshl_add_u64 v[102:103], v[100:101], 2, v[104:105]\n v_lshl_add_u64 v[100:101], v[102:103], 2, v[106:107]\n v_lshl_add_u64 v[102:103], v[100:101], 2, v[106:107]\n")
DEFINE_KERNEL(k_cmpu64, "v_cmp_lt_u64_e32 vcc, v[100:101], v[102:103]\n v_cmp_lt_u64_e32 vcc, v[104:105], v[106:107]\n v_cmp_lt_u64_e32 vcc, v[100:101], v[102:103]\n v_cmp_lt_u64_e32 vcc, v[104:105], v[106:107]\n")
K4_3(k_bfi, "v_bfi_b32")
K4(k_mulhi, "v_mul_hi_u32")
K4(k_max3u, "v_max_i32_e32")

typedef void (*kern_t)(int, unsigned long long*, float*);
struct Entry { const char* name; kern_t k; };

int main() {
  const Entry es[] = {{"v_fma_f32", k_fma}, {"v_mul_f32_e32", k_mul}, {"v_cvt_f32_ubyte1", k_cvtub}, {"v_max3_f32", k_max3}, {"v_min_f32_e32", k_min},
                      {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64}, {"v_cndmask_b32_e32 (vcc)", k_cnd32}, {"v_cmp_le_f32_e64 -> sgpr", k_cmp64},
                      {"v_cmp_lt_u32_e32 -> vcc", k_cmp32}, {"v_and_b32_e32", k_and}, {"v_lshl_or_b32", k_lshlor}, {"v_min_u32_e32", k_minu},
                      {"v_mov_b32_e32", k_mov}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"s_and/or_b64", k_sand},
                      {"mix fma/cndmask/cvt/cmp", k_mix},
                      {"v_add_u32_e32", k_addu}, {"v_sub_f32_e32", k_subf}, {"v_add_f32_e32", k_addf}, {"v_max_f32_e32", k_maxf}, {"v_or_b32_e32", k_or},
                      {"v_xor_b32_e32", k_xor}, {"v_lshlrev_b32_e32", k_shl}, {"v_lshrrev_b32_e32", k_shr}, {"v_ashrrev_i32_e32", k_ashr},
                      {"v_mul_u32_u24_e32", k_mul24}, {"v_max_u32_e32", k_maxu}, {"v_min_i32_e32", k_mini}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm},
                      {"v_and_or_b32", k_andor}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd}, {"v_mad_u32_u24", k_mad24}, {"v_med3_f32", k_med3},
                      {"v_min3_u32", k_min3u}, {"v_alignbit_b32", k_alignbit}, {"v_cvt_f32_u32_e32", k_cvtu32}, {"v_cvt_f32_ubyte0_e32", k_cvtub0},
                      {"v_rcp_f32_e32", k_rcp}, {"v_cvt_f32_f16_e32", k_cvtf16}, {"v_cvt_f32_u32_sdwa BYTE_n", k_cvt_sdwa}, {"v_mul_f32_sdwa", k_mul_sdwa},
                      {"v_fmac_f32_e32", k_fmac}, {"v_fmamk_f32", k_fmamk}, {"v_cmp_le_f32_e32 -> vcc", k_cmp_f_e32}, {"cmp_e32 + cndmask_e32 pairs", k_cnd_vccw},
                      {"ds_write_b32 (lane-private)", k_dswrite}, {"ds_read_b32 (lane-private)", k_dsread},
                      {"v_bitop3_b32", k_bitop3}, {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbl_b32", k_ffbl}, {"v_bfm_b32", k_bfm}, {"v_add_lshl_u32", k_addlshl},
                      {"v_fma_mix_f32 (f32 srcs)", k_fmamix}, {"v_fma_mix_f32 (f16 src0 hi/lo)", k_fmamix_hi}, {"v_cvt_pk_f32_fp8", k_cvtpkfp8},
                      {"v_pk_fma_f16", k_pkfmaf16}, {"v_pk_max_f16", k_pkmaxf16}, {"v_pk_min_f16", k_pkminf16}, {"v_mul_lo_u32", k_mullo},
                      {"v_dot2_f32_f16", k_dot2}, {"v_not_b32", k_not}, {"ds_read_u8 (lane-private)", k_dsread_u8}, {"ds_read_u16 (lane-private)", k_dsread_u16},
                      {"ds_write_b64 (lane-private)", k_dswrite_b64}, {"ds_read_b64 (lane-private)", k_dsread_b64},
                      {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas}, {"v_div_fixup_f32", k_divfixup}, {"v_sqrt_f32", k_sqrt}, {"v_rsq_f32", k_rsq},
                      {"v_cmp_class_f32 -> vcc", k_cmpclass}, {"v_rndne_f32", k_rndne}, {"v_cvt_i32_f32", k_cvti32}, {"v_cvt_u32_f32", k_cvtu32f},
                      {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_ldexp_f32", k_ldexp}, {"v_min3_f32", k_min3f}, {"v_cvt_scalef32_pk_f32_fp8", k_cvtscalefp8},
                      {"v_lshl_add_u64", k_lshladd64}, {"v_cmp_lt_u64 -> vcc", k_cmpu64}, {"v_bfi_b32", k_bfi}, {"v_mul_hi_u32", k_mulhi}, {"v_max_i32", k_max3u}};
  const int loops = 20000;
  unsigned long long* out; float* sink;
  CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&sink, 4));
  printf("%-32s %8s %8s %8s %8s   (shader cycles per instruction per SIMD)\n", "instruction", "1 w/SIMD", "2", "4", "8");
  for (const Entry& e : es) {
    printf("%-32s", e.name);
    fflush(stdout);
    float wall8 = 0;
    for (int w = 1; w <= 8; w *= 2) {
      // one block per CU holding w waves per SIMD (w = 8: two 1024-thread blocks per CU)
      const int threads = w <= 4 ? 256 * w : 1024;
      const int blocks = w <= 4 ? 256 : 512;
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, (e.name[0] == 'd' ? (strstr(e.name, "b64") ? -loops - (1 << 24) : -loops) : loops), out, sink);
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, (e.name[0] == 'd' ? (strstr(e.name, "b64") ? -loops - (1 << 24) : -loops) : loops), out, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (w == 8) wall8 = ms;
      std::vector<unsigned long long> h(blocks * threads / 64);
      CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      // every wave runs loops*32 instructions; w waves share a SIMD
      printf(" %8.2f", med / (loops * 32.0) / w);
    }
    printf("   | 8 w/SIMD by wall clock: %.2f cyc at 2.35 GHz\n", wall8 * 1e-3 * 2.35e9 / (loops * 32.0) / 8);
    fflush(stdout);
  }
  return 0;
}
