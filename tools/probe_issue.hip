// probe_issue.hip — issue cost (shader cycles per wave-instruction per SIMD) of the opcodes the lean Viterbi step uses,
// and of the candidates that could replace them, on gfx950.  Each probe is an asm body of NI instructions over 8
// independent register chains ("indep") or one dependent chain ("dep"), repeated REP x iters times by W waves per SIMD
// (W = 1, 2, 4, 8; every CU gets 4 W waves, one workgroup of 256 W threads, or two of 1024 for W = 8).  Reported:
//   cyc/instr = median over waves of (s_memtime delta) / (W * instructions per wave)
// i.e. the SIMD's issue time per wave-instruction when W waves share it.  The same number from the wall clock
// (event time x measured shader clock / instructions per SIMD) is printed beside it as a cross-check.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_issue.bin tools/probe_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP 8
struct Out { unsigned long long cyc, wall; };

#define PROLOGUE                                                                                                      \
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,      \
        a7 = a0 + 7;                                                                                                  \
  float b0 = 1.0001f, b1 = 0.5f, b2 = 0.25f, b3 = 0.125f;                                                             \
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q0 = {b0, b1}, q1 = {b2, b3};                       \
  unsigned u0 = threadIdx.x, u1 = 0x380;                                                                              \
  float s0 = seed[0], s1 = seed[1];                                                                                   \
  v2f sp = {seed[2], seed[3]};                                                                                        \
  (void)s0; (void)s1; (void)sp;                                                                                       \
  __syncthreads();                                                                                                    \
  const unsigned long long t0 = clock64(), w0 = wall_clock64();

#define EPILOGUE                                                                                                      \
  const unsigned long long t1 = clock64(), w1 = wall_clock64();                                                      \
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + p0.x +   \
                                                  p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + q0.x + q1.y +      \
                                                  __uint_as_float(u0);                                               \
  if ((threadIdx.x & 63) == 0) {                                                                                      \
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                                       \
    out[w].cyc = t1 - t0; out[w].wall = w1 - w0;                                                                      \
  }

#define OPS                                                                                                           \
  [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),           \
      [a7] "+v"(a7), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3), [p0] "+v"(p0), [p1] "+v"(p1),       \
      [p2] "+v"(p2), [p3] "+v"(p3), [q0] "+v"(q0), [q1] "+v"(q1), [u0] "+v"(u0), [u1] "+v"(u1)
#define INS [s0] "s"(s0), [s1] "s"(s1), [sp] "s"(sp)

#define PROBE(NAME, BODY)                                                                                             \
  __global__ __launch_bounds__(1024) void NAME(float* sink, Out* out, const float* seed, int iters)                  \
  {                                                                                                                   \
    extern __shared__ char lds_pad[];                                                                                 \
    (void)lds_pad;                                                                                                    \
    PROLOGUE                                                                                                          \
    for (int i = 0; i < iters; i++) {                                                                                 \
      _Pragma("unroll") for (int r = 0; r < REP; r++) asm volatile(BODY : OPS : INS : "vcc", "s40", "s41");            \
    }                                                                                                                 \
    EPILOGUE                                                                                                          \
  }

// eight instructions over eight chains
#define X8(OP, TAIL)                                                                                                  \
  OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a1], %[a1]" TAIL "\n\t" OP " %[a2], %[a2]" TAIL "\n\t" OP " %[a3], %[a3]" TAIL "\n\t" \
  OP " %[a4], %[a4]" TAIL "\n\t" OP " %[a5], %[a5]" TAIL "\n\t" OP " %[a6], %[a6]" TAIL "\n\t" OP " %[a7], %[a7]" TAIL
// eight instructions on one chain
#define D8(OP, TAIL)                                                                                                  \
  OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a0], %[a0]" TAIL "\n\t" \
  OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a0], %[a0]" TAIL "\n\t" OP " %[a0], %[a0]" TAIL
// packed: four chains x 2
#define P8(OP, TAIL)                                                                                                  \
  OP " %[p0], %[p0]" TAIL "\n\t" OP " %[p1], %[p1]" TAIL "\n\t" OP " %[p2], %[p2]" TAIL "\n\t" OP " %[p3], %[p3]" TAIL "\n\t" \
  OP " %[p0], %[p0]" TAIL "\n\t" OP " %[p1], %[p1]" TAIL "\n\t" OP " %[p2], %[p2]" TAIL "\n\t" OP " %[p3], %[p3]" TAIL

#define DPPQ " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define DPPR " row_ror:8 row_mask:0xf bank_mask:0xf"
#define DPPH " row_half_mirror row_mask:0xf bank_mask:0xf"
#define DPPS " row_shr:1 row_mask:0xf bank_mask:0xf"

PROBE(p_fma, X8("v_fma_f32", ", %[b0], %[b1]"))
PROBE(p_fma_dep, D8("v_fma_f32", ", %[b0], %[b1]"))
PROBE(p_fma_sgpr, X8("v_fma_f32", ", %[s0], %[b1]"))
PROBE(p_add, X8("v_add_f32", ", %[b0]"))
PROBE(p_add_dep, D8("v_add_f32", ", %[b0]"))
PROBE(p_max, X8("v_max_f32", ", %[b0]"))
PROBE(p_max3, X8("v_max3_f32", ", %[b0], %[b1]"))
PROBE(p_mov, "v_mov_b32 %[a0], %[a1]\n\tv_mov_b32 %[a1], %[a2]\n\tv_mov_b32 %[a2], %[a3]\n\tv_mov_b32 %[a3], %[a4]\n\t"
             "v_mov_b32 %[a4], %[a5]\n\tv_mov_b32 %[a5], %[a6]\n\tv_mov_b32 %[a6], %[a7]\n\tv_mov_b32 %[a7], %[a0]")
PROBE(p_addu, "v_add_u32 %[a0], %[a0], %[u0]\n\tv_add_u32 %[a1], %[a1], %[u0]\n\tv_add_u32 %[a2], %[a2], %[u0]\n\tv_add_u32 %[a3], %[a3], %[u0]\n\t"
              "v_add_u32 %[a4], %[a4], %[u0]\n\tv_add_u32 %[a5], %[a5], %[u0]\n\tv_add_u32 %[a6], %[a6], %[u0]\n\tv_add_u32 %[a7], %[a7], %[u0]")
PROBE(p_lshl_or, X8("v_lshl_or_b32", ", 1, %[u0]"))
PROBE(p_pk_fma, P8("v_pk_fma_f32", ", %[q0], %[q1]"))
PROBE(p_pk_fma_sgpr, P8("v_pk_fma_f32", ", %[sp], %[q1] op_sel_hi:[1,0,1]"))
PROBE(p_pk_add, P8("v_pk_add_f32", ", %[q0]"))
PROBE(p_pk_mul, P8("v_pk_mul_f32", ", %[q0]"))
PROBE(p_pk_add_dep, "v_pk_add_f32 %[p0], %[p0], %[q0]\n\tv_pk_add_f32 %[p0], %[p0], %[q0]\n\tv_pk_add_f32 %[p0], %[p0], %[q0]\n\tv_pk_add_f32 %[p0], %[p0], %[q0]\n\t"
                    "v_pk_add_f32 %[p0], %[p0], %[q0]\n\tv_pk_add_f32 %[p0], %[p0], %[q0]\n\tv_pk_add_f32 %[p0], %[p0], %[q0]\n\tv_pk_add_f32 %[p0], %[p0], %[q0]")
// DPP as an operand modifier (sources are not written inside the body: no DPP read-after-VALU-write hazard)
#define DPP8(OP, CTRL)                                                                                                \
  OP " %[a0], %[b0], %[b1]" CTRL "\n\t" OP " %[a1], %[b0], %[b1]" CTRL "\n\t" OP " %[a2], %[b1], %[b2]" CTRL "\n\t"   \
  OP " %[a3], %[b2], %[b3]" CTRL "\n\t" OP " %[a4], %[b0], %[b3]" CTRL "\n\t" OP " %[a5], %[b1], %[b0]" CTRL "\n\t"   \
  OP " %[a6], %[b2], %[b1]" CTRL "\n\t" OP " %[a7], %[b3], %[b2]" CTRL
PROBE(p_sub_dpp_quad, DPP8("v_sub_f32_dpp", DPPQ))
PROBE(p_max_dpp_quad, DPP8("v_max_f32_dpp", DPPQ))
PROBE(p_max_dpp_ror8, DPP8("v_max_f32_dpp", DPPR))
PROBE(p_max_dpp_hmir, DPP8("v_max_f32_dpp", DPPH))
PROBE(p_add_dpp_shr1, DPP8("v_add_f32_dpp", DPPS))
PROBE(p_mov_dpp_quad, "v_mov_b32_dpp %[a0], %[b0]" DPPQ "\n\tv_mov_b32_dpp %[a1], %[b1]" DPPQ "\n\tv_mov_b32_dpp %[a2], %[b2]" DPPQ "\n\tv_mov_b32_dpp %[a3], %[b3]" DPPQ "\n\t"
                      "v_mov_b32_dpp %[a4], %[b0]" DPPQ "\n\tv_mov_b32_dpp %[a5], %[b1]" DPPQ "\n\tv_mov_b32_dpp %[a6], %[b2]" DPPQ "\n\tv_mov_b32_dpp %[a7], %[b3]" DPPQ)
// dependent DPP chain with the required 2 wait states (s_nop 1) before each DPP read of a fresh VALU result
PROBE(p_max_dpp_dep, "s_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\ts_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\ts_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\ts_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ "\n\ts_nop 1\n\tv_max_f32_dpp %[a0], %[a0], %[b0]" DPPQ)
PROBE(p_cmpclass_addc, "v_cmp_class_f32_e32 vcc, %[a0], %[u1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                       "v_cmp_class_f32_e32 vcc, %[a1], %[u1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                       "v_cmp_class_f32_e32 vcc, %[a2], %[u1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                       "v_cmp_class_f32_e32 vcc, %[a3], %[u1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc")
PROBE(p_cmpgt_addc, "v_cmp_gt_f32_e32 vcc, %[a0], %[a1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                    "v_cmp_gt_f32_e32 vcc, %[a1], %[a2]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                    "v_cmp_gt_f32_e32 vcc, %[a2], %[a3]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                    "v_cmp_gt_f32_e32 vcc, %[a3], %[a4]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc")
PROBE(p_cmpclass, "v_cmp_class_f32_e32 vcc, %[a0], %[u1]\n\tv_cmp_class_f32_e32 vcc, %[a1], %[u1]\n\tv_cmp_class_f32_e32 vcc, %[a2], %[u1]\n\tv_cmp_class_f32_e32 vcc, %[a3], %[u1]\n\t"
                  "v_cmp_class_f32_e32 vcc, %[a4], %[u1]\n\tv_cmp_class_f32_e32 vcc, %[a5], %[u1]\n\tv_cmp_class_f32_e32 vcc, %[a6], %[u1]\n\tv_cmp_class_f32_e32 vcc, %[a7], %[u1]")
PROBE(p_cmpgt, "v_cmp_gt_f32_e32 vcc, %[a0], %[a1]\n\tv_cmp_gt_f32_e32 vcc, %[a1], %[a2]\n\tv_cmp_gt_f32_e32 vcc, %[a2], %[a3]\n\tv_cmp_gt_f32_e32 vcc, %[a3], %[a4]\n\t"
               "v_cmp_gt_f32_e32 vcc, %[a4], %[a5]\n\tv_cmp_gt_f32_e32 vcc, %[a5], %[a6]\n\tv_cmp_gt_f32_e32 vcc, %[a6], %[a7]\n\tv_cmp_gt_f32_e32 vcc, %[a7], %[a0]")
PROBE(p_cmpgt_e64, "v_cmp_gt_f32_e64 s[40:41], %[a0], %[a1]\n\tv_cmp_gt_f32_e64 s[40:41], %[a1], %[a2]\n\tv_cmp_gt_f32_e64 s[40:41], %[a2], %[a3]\n\tv_cmp_gt_f32_e64 s[40:41], %[a3], %[a4]\n\t"
                   "v_cmp_gt_f32_e64 s[40:41], %[a4], %[a5]\n\tv_cmp_gt_f32_e64 s[40:41], %[a5], %[a6]\n\tv_cmp_gt_f32_e64 s[40:41], %[a6], %[a7]\n\tv_cmp_gt_f32_e64 s[40:41], %[a7], %[a0]")
PROBE(p_addc, "v_addc_co_u32_e32 %[a0], vcc, %[a0], %[a0], vcc\n\tv_addc_co_u32_e32 %[a1], vcc, %[a1], %[a1], vcc\n\tv_addc_co_u32_e32 %[a2], vcc, %[a2], %[a2], vcc\n\tv_addc_co_u32_e32 %[a3], vcc, %[a3], %[a3], vcc\n\t"
              "v_addc_co_u32_e32 %[a4], vcc, %[a4], %[a4], vcc\n\tv_addc_co_u32_e32 %[a5], vcc, %[a5], %[a5], vcc\n\tv_addc_co_u32_e32 %[a6], vcc, %[a6], %[a6], vcc\n\tv_addc_co_u32_e32 %[a7], vcc, %[a7], %[a7], vcc")
PROBE(p_cndmask, X8("v_cndmask_b32", ", %[b0], vcc"))
PROBE(p_swap16, "v_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a2], %[a3]\n\tv_permlane16_swap_b32 %[a4], %[a5]\n\tv_permlane16_swap_b32 %[a6], %[a7]\n\t"
                "v_permlane16_swap_b32 %[b0], %[b1]\n\tv_permlane16_swap_b32 %[b2], %[b3]\n\tv_permlane16_swap_b32 %[a0], %[a2]\n\tv_permlane16_swap_b32 %[a4], %[a6]")
PROBE(p_swap32, "v_permlane32_swap_b32 %[a0], %[a1]\n\tv_permlane32_swap_b32 %[a2], %[a3]\n\tv_permlane32_swap_b32 %[a4], %[a5]\n\tv_permlane32_swap_b32 %[a6], %[a7]\n\t"
                "v_permlane32_swap_b32 %[b0], %[b1]\n\tv_permlane32_swap_b32 %[b2], %[b3]\n\tv_permlane32_swap_b32 %[a0], %[a2]\n\tv_permlane32_swap_b32 %[a4], %[a6]")
PROBE(p_swap16_dep, "v_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a0], %[a1]\n\t"
                    "v_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a0], %[a1]\n\tv_permlane16_swap_b32 %[a0], %[a1]")
PROBE(p_snop1_fma, "s_nop 1\n\tv_fma_f32 %[a0], %[a0], %[b0], %[b1]\n\ts_nop 1\n\tv_fma_f32 %[a1], %[a1], %[b0], %[b1]\n\ts_nop 1\n\tv_fma_f32 %[a2], %[a2], %[b0], %[b1]\n\t"
                   "s_nop 1\n\tv_fma_f32 %[a3], %[a3], %[b0], %[b1]")       // counted as 4 instructions: the extra over p_fma is the s_nop
PROBE(p_smov_fma, "s_mov_b32 s40, s41\n\tv_fma_f32 %[a0], %[a0], %[b0], %[b1]\n\ts_mov_b32 s40, s41\n\tv_fma_f32 %[a1], %[a1], %[b0], %[b1]\n\ts_mov_b32 s40, s41\n\tv_fma_f32 %[a2], %[a2], %[b0], %[b1]\n\t"
                  "s_mov_b32 s40, s41\n\tv_fma_f32 %[a3], %[a3], %[b0], %[b1]")
PROBE(p_bpermute, "ds_bpermute_b32 %[a0], %[u0], %[a0]\n\tds_bpermute_b32 %[a1], %[u0], %[a1]\n\tds_bpermute_b32 %[a2], %[u0], %[a2]\n\tds_bpermute_b32 %[a3], %[u0], %[a3]\n\t"
                  "ds_bpermute_b32 %[a4], %[u0], %[a4]\n\tds_bpermute_b32 %[a5], %[u0], %[a5]\n\tds_bpermute_b32 %[a6], %[u0], %[a6]\n\tds_bpermute_b32 %[a7], %[u0], %[a7]\n\ts_waitcnt lgkmcnt(0)")
PROBE(p_swizzle, "ds_swizzle_b32 %[a0], %[a0] offset:0x401f\n\tds_swizzle_b32 %[a1], %[a1] offset:0x401f\n\tds_swizzle_b32 %[a2], %[a2] offset:0x401f\n\tds_swizzle_b32 %[a3], %[a3] offset:0x401f\n\t"
                 "ds_swizzle_b32 %[a4], %[a4] offset:0x401f\n\tds_swizzle_b32 %[a5], %[a5] offset:0x401f\n\tds_swizzle_b32 %[a6], %[a6] offset:0x401f\n\tds_swizzle_b32 %[a7], %[a7] offset:0x401f\n\ts_waitcnt lgkmcnt(0)")

// ---- whole trellis steps, as the lean kernel issues them (dependent through a0 = path metric) ---------------------
// (explicit registers v[100:107]: the packed result pair is read as two 32-bit DPP operands)
// current q=0 step with one LLR: pk_fma(keep,send) ; [decision of the previous step] ; sub_dpp ; max_dpp       = 5 VALU
#define HARD "vcc", "s40", "s41", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109"
#define PROBE_HARD(NAME, BODY)                                                                                        \
  __global__ __launch_bounds__(1024) void NAME(float* sink, Out* out, const float* seed, int iters)                  \
  {                                                                                                                   \
    extern __shared__ char lds_pad[];                                                                                 \
    (void)lds_pad;                                                                                                    \
    PROLOGUE                                                                                                          \
    asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v104, 0\n\t" \
                 "v_mov_b32 v105, 0\n\tv_mov_b32 v106, 1.0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, %2"       \
                 ::"v"(a0), "v"(a1), "v"(((threadIdx.x ^ 32u) & 63u) * 4u) : HARD);                                    \
    for (int i = 0; i < iters; i++) {                                                                                 \
      _Pragma("unroll") for (int r = 0; r < REP; r++) asm volatile(BODY : [u0] "+v"(u0) : [u1] "v"(u1), INS : HARD);  \
    }                                                                                                                 \
    asm volatile("v_mov_b32 %0, v102" : "=v"(a0) :: HARD);                                                            \
    EPILOGUE                                                                                                          \
  }
// v[102:103] = {pre, pre-copy}, v[106:107] = lane coefficients {ka, sa}, v[100:101] = {keep, send}, v104 = d
PROBE_HARD(p_step_cur, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                       "v_cmp_class_f32_e32 vcc, v104, %[u1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                       "v_sub_f32_dpp v104, v101, v100" DPPQ "\n\tv_max_f32_dpp v102, v101, v100" DPPQ)
// the same with two plain fmas instead of the packed one                                                       = 6 VALU
PROBE_HARD(p_step_2fma, "v_fma_f32 v100, v106, %[s0], v102\n\tv_fma_f32 v101, v107, %[s0], v102\n\t"
                        "v_cmp_class_f32_e32 vcc, v104, %[u1]\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                        "v_sub_f32_dpp v104, v101, v100" DPPQ "\n\tv_max_f32_dpp v102, v101, v100" DPPQ)
// swap step (q = 4): pk_fma ; cmp ; addc ; swap ; max                                                          = 5 VALU
PROBE_HARD(p_step_swap, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                        "v_cmp_gt_f32_e32 vcc, v104, v105\n\tv_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                        "v_permlane16_swap_b32 v100, v101\n\tv_max_f32 v102, v100, v101")
// metric-only chain of the DPP step: fma, fma, (2 wait states filled by the hardware or not), max_dpp
PROBE_HARD(p_chain_dpp, "v_fma_f32 v100, v106, %[s0], v102\n\tv_fma_f32 v101, v107, %[s0], v102\n\ts_nop 1\n\t"
                        "v_max_f32_dpp v102, v101, v100" DPPQ)


// ---- second set: which opcodes issue in 2 cycles (candidates for a cheaper step) -----------------------------------
PROBE(p_sub, X8("v_sub_f32", ", %[b0]"))
PROBE(p_mul, X8("v_mul_f32", ", %[b0]"))
PROBE(p_min, X8("v_min_f32", ", %[b0]"))
PROBE(p_add_sgpr, X8("v_add_f32", ", %[s0]"))
PROBE(p_add_e64, X8("v_add_f32_e64", ", %[b0]"))
PROBE(p_add_neg, X8("v_add_f32_e64", ", -%[b0]"))
PROBE(p_fmac, "v_fmac_f32 %[a0], %[b0], %[b1]\n\tv_fmac_f32 %[a1], %[b0], %[b1]\n\tv_fmac_f32 %[a2], %[b0], %[b1]\n\tv_fmac_f32 %[a3], %[b0], %[b1]\n\t"
              "v_fmac_f32 %[a4], %[b0], %[b1]\n\tv_fmac_f32 %[a5], %[b0], %[b1]\n\tv_fmac_f32 %[a6], %[b0], %[b1]\n\tv_fmac_f32 %[a7], %[b0], %[b1]")
PROBE(p_fma_c0, X8("v_fma_f32", ", %[b0], 0"))
PROBE(p_fma_aab, X8("v_fma_f32", ", %[b0], %[b0]"))
PROBE(p_and, X8("v_and_b32", ", %[u0]"))
PROBE(p_or, X8("v_or_b32", ", %[u0]"))
PROBE(p_xor, X8("v_xor_b32", ", %[u0]"))
PROBE(p_lshlrev, "v_lshlrev_b32 %[a0], 1, %[a0]\n\tv_lshlrev_b32 %[a1], 1, %[a1]\n\tv_lshlrev_b32 %[a2], 1, %[a2]\n\tv_lshlrev_b32 %[a3], 1, %[a3]\n\t"
                 "v_lshlrev_b32 %[a4], 1, %[a4]\n\tv_lshlrev_b32 %[a5], 1, %[a5]\n\tv_lshlrev_b32 %[a6], 1, %[a6]\n\tv_lshlrev_b32 %[a7], 1, %[a7]")
PROBE(p_lshrrev, "v_lshrrev_b32 %[a0], 31, %[a0]\n\tv_lshrrev_b32 %[a1], 31, %[a1]\n\tv_lshrrev_b32 %[a2], 31, %[a2]\n\tv_lshrrev_b32 %[a3], 31, %[a3]\n\t"
                 "v_lshrrev_b32 %[a4], 31, %[a4]\n\tv_lshrrev_b32 %[a5], 31, %[a5]\n\tv_lshrrev_b32 %[a6], 31, %[a6]\n\tv_lshrrev_b32 %[a7], 31, %[a7]")
PROBE(p_alignbit, X8("v_alignbit_b32", ", %[b0], 31"))
PROBE(p_bfi, X8("v_bfi_b32", ", %[b0], %[b1]"))
PROBE(p_perm, X8("v_perm_b32", ", %[b0], %[u0]"))
PROBE(p_max_i32, X8("v_max_i32", ", %[b0]"))
PROBE(p_max_u32, X8("v_max_u32", ", %[b0]"))
PROBE(p_max_f16, X8("v_max_f16", ", %[b0]"))
PROBE(p_pk_max_f16, X8("v_pk_max_f16", ", %[b0]"))
PROBE(p_add_co, "v_add_co_u32 %[a0], vcc, %[a0], %[a0]\n\tv_add_co_u32 %[a1], vcc, %[a1], %[a1]\n\tv_add_co_u32 %[a2], vcc, %[a2], %[a2]\n\tv_add_co_u32 %[a3], vcc, %[a3], %[a3]\n\t"
                "v_add_co_u32 %[a4], vcc, %[a4], %[a4]\n\tv_add_co_u32 %[a5], vcc, %[a5], %[a5]\n\tv_add_co_u32 %[a6], vcc, %[a6], %[a6]\n\tv_add_co_u32 %[a7], vcc, %[a7], %[a7]")
PROBE(p_cmp_cndmask, "v_cmp_gt_f32_e32 vcc, %[a0], %[a1]\n\tv_cndmask_b32 %[a2], %[a2], %[b0], vcc\n\tv_cmp_gt_f32_e32 vcc, %[a1], %[a3]\n\tv_cndmask_b32 %[a4], %[a4], %[b0], vcc\n\t"
                     "v_cmp_gt_f32_e32 vcc, %[a3], %[a5]\n\tv_cndmask_b32 %[a6], %[a6], %[b0], vcc\n\tv_cmp_gt_f32_e32 vcc, %[a5], %[a7]\n\tv_cndmask_b32 %[a0], %[a0], %[b0], vcc")
PROBE(p_cndmask_e64, "s_mov_b64 s[40:41], 0x5555\n\tv_cndmask_b32_e64 %[a0], %[a0], %[b0], s[40:41]\n\tv_cndmask_b32_e64 %[a1], %[a1], %[b0], s[40:41]\n\tv_cndmask_b32_e64 %[a2], %[a2], %[b0], s[40:41]\n\t"
                     "v_cndmask_b32_e64 %[a3], %[a3], %[b0], s[40:41]\n\tv_cndmask_b32_e64 %[a4], %[a4], %[b0], s[40:41]\n\tv_cndmask_b32_e64 %[a5], %[a5], %[b0], s[40:41]\n\t"
                     "v_cndmask_b32_e64 %[a6], %[a6], %[b0], s[40:41]\n\tv_cndmask_b32_e64 %[a7], %[a7], %[b0], s[40:41]")
PROBE(p_cndmask_b, "v_cndmask_b32 %[a0], %[b1], %[b0], vcc\n\tv_cndmask_b32 %[a1], %[b1], %[b0], vcc\n\tv_cndmask_b32 %[a2], %[b1], %[b0], vcc\n\tv_cndmask_b32 %[a3], %[b1], %[b0], vcc\n\t"
                   "v_cndmask_b32 %[a4], %[b1], %[b0], vcc\n\tv_cndmask_b32 %[a5], %[b1], %[b0], vcc\n\tv_cndmask_b32 %[a6], %[b1], %[b0], vcc\n\tv_cndmask_b32 %[a7], %[b1], %[b0], vcc")
PROBE(p_max_noieee, "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n\t" X8("v_max_f32", ", %[b0]") "\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 1")
PROBE(p_add_f64, P8("v_add_f64", ", %[q0]"))
PROBE(p_fma_f64, P8("v_fma_f64", ", %[q0], %[q1]"))
PROBE(p_max_f64, P8("v_max_f64", ", %[q0]"))
PROBE(p_mul_f64, P8("v_mul_f64", ", %[q0]"))
PROBE(p_pk_mov, P8("v_pk_mov_b32", ", %[q0]"))
PROBE(p_writelane, "v_writelane_b32 %[a0], s40, 3\n\tv_writelane_b32 %[a1], s40, 4\n\tv_writelane_b32 %[a2], s40, 5\n\tv_writelane_b32 %[a3], s40, 6\n\t"
                   "v_writelane_b32 %[a4], s40, 7\n\tv_writelane_b32 %[a5], s40, 8\n\tv_writelane_b32 %[a6], s40, 9\n\tv_writelane_b32 %[a7], s40, 10")
PROBE(p_mov_sgpr, "v_mov_b32 %[a0], %[s0]\n\tv_mov_b32 %[a1], %[s0]\n\tv_mov_b32 %[a2], %[s0]\n\tv_mov_b32 %[a3], %[s0]\n\t"
                  "v_mov_b32 %[a4], %[s0]\n\tv_mov_b32 %[a5], %[s0]\n\tv_mov_b32 %[a6], %[s0]\n\tv_mov_b32 %[a7], %[s0]")
PROBE(p_add_sdwa, X8("v_add_f32_sdwa", ", %[b0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"))
PROBE(p_sub_dpp_b0, "v_sub_f32_dpp %[a0], %[b0], %[b1]" DPPQ "\n\tv_sub_f32 %[a1], %[b0], %[b1]\n\tv_sub_f32_dpp %[a2], %[b1], %[b2]" DPPQ "\n\tv_sub_f32 %[a3], %[b0], %[b1]\n\t"
                    "v_sub_f32_dpp %[a4], %[b0], %[b3]" DPPQ "\n\tv_sub_f32 %[a5], %[b0], %[b1]\n\tv_sub_f32_dpp %[a6], %[b2], %[b1]" DPPQ "\n\tv_sub_f32 %[a7], %[b0], %[b1]")
// LDS broadcast read (all lanes the same address): VALU-free delivery of uniform values into VGPRs
PROBE(p_ldsbc_fma, "ds_read_b128 v[100:103], %[u1]\n\t" X8("v_fma_f32", ", %[b0], %[b1]") "\n\ts_waitcnt lgkmcnt(0)")
// candidate new steps (explicit registers as above; v105 = sigma', v108 = acc)
// DPP step with the sign trick: pk_fma ; sub_dpp ; max_dpp ; e = fma(sigma', d, +0) ; acc = alignbit(acc, e, 31)
PROBE_HARD(p_step_sign, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                        "v_fma_f32 v105, v104, v106, 0\n\tv_alignbit_b32 %[u0], %[u0], v105, 31\n\t"
                        "v_sub_f32_dpp v104, v101, v100" DPPQ "\n\tv_max_f32_dpp v102, v101, v100" DPPQ)
// swap step with the sign trick: pk_fma ; swap ; max ; e = a0 - a1 ; alignbit
PROBE_HARD(p_step_swap_sign, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                             "v_sub_f32 v105, v104, v103\n\tv_alignbit_b32 %[u0], %[u0], v105, 31\n\t"
                             "v_permlane16_swap_b32 v100, v101\n\tv_max_f32 v102, v100, v101")

// ---- third set: integer-compare step (all candidates positive: unsigned order of the bit patterns = float order) and LDS-pipe
// exchanges for the lane bits 4 / 5.  v108 = d (integer difference, unused), v109 = byte address of lane ^ 32 for ds_bpermute.
// DPP step: pk_fma ; addc (records the previous decision, its carry-out = the per-lane tie rule of this step) ; subb_dpp ; max_u32_dpp
PROBE_HARD(p_step_int, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                       "v_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\ts_nop 0\n\t"
                       "v_subb_co_u32_dpp v108, vcc, v101, v100, vcc" DPPQ "\n\tv_max_u32_dpp v102, v101, v100" DPPQ)
// the same with v_cndmask_b32_dpp as the selector
PROBE_HARD(p_step_int_cnd, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                           "v_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\ts_nop 0\n\t"
                           "v_subb_co_u32_dpp v108, vcc, v101, v100, vcc" DPPQ "\n\tv_cndmask_b32_dpp v102, v101, v100, vcc" DPPQ)
// lane bit 4 through ds_swizzle (LDS pipe, no VALU slot), float compare: pk_fma ; swizzle ; cmp_class ; addc ; wait ; sub ; max
PROBE_HARD(p_step_swz, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                       "v_cmp_class_f32_e32 vcc, v104, %[u1]\n\tds_swizzle_b32 v105, v101 offset:0x401f\n\t"
                       "v_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\ts_waitcnt lgkmcnt(0)\n\t"
                       "v_sub_f32 v104, v105, v100\n\tv_max_f32 v102, v105, v100")
// lane bit 4 through ds_swizzle, integer compare: pk_fma ; swizzle ; addc ; wait ; subb ; max_u32
PROBE_HARD(p_step_swz_int, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                           "ds_swizzle_b32 v105, v101 offset:0x401f\n\t"
                           "v_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\ts_waitcnt lgkmcnt(0)\n\t"
                           "v_subb_co_u32_e32 v108, vcc, v105, v100, vcc\n\tv_max_u32 v102, v105, v100")
// lane bit 5 through ds_bpermute, integer compare
PROBE_HARD(p_step_bperm_int, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                             "ds_bpermute_b32 v105, v109, v101\n\t"
                             "v_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "v_subb_co_u32_e32 v108, vcc, v105, v100, vcc\n\tv_max_u32 v102, v105, v100")
// swap step, integer compare: pk_fma ; swap ; addc ; sub_co ; max
PROBE_HARD(p_step_swap_int, "v_pk_fma_f32 v[100:101], v[106:107], %[sp], v[102:103] op_sel_hi:[1,0,0]\n\t"
                            "v_addc_co_u32_e32 %[u0], vcc, %[u0], %[u0], vcc\n\t"
                            "v_permlane16_swap_b32 v100, v101\n\tv_sub_co_u32_e32 v108, vcc, v100, v101\n\tv_max_u32 v102, v100, v101")
PROBE(p_subb_dpp, "v_subb_co_u32_dpp %[a0], vcc, %[b0], %[b1], vcc" DPPQ "\n\tv_subb_co_u32_dpp %[a1], vcc, %[b1], %[b2], vcc" DPPQ "\n\t"
                  "v_subb_co_u32_dpp %[a2], vcc, %[b2], %[b3], vcc" DPPQ "\n\tv_subb_co_u32_dpp %[a3], vcc, %[b3], %[b0], vcc" DPPQ "\n\t"
                  "v_subb_co_u32_dpp %[a4], vcc, %[b0], %[b2], vcc" DPPQ "\n\tv_subb_co_u32_dpp %[a5], vcc, %[b1], %[b3], vcc" DPPQ "\n\t"
                  "v_subb_co_u32_dpp %[a6], vcc, %[b2], %[b0], vcc" DPPQ "\n\tv_subb_co_u32_dpp %[a7], vcc, %[b3], %[b1], vcc" DPPQ)
PROBE(p_max_u32_dpp, DPP8("v_max_u32_dpp", DPPQ))

struct Probe { const char* name; void (*fn)(float*, Out*, const float*, int); int ni; const char* what; };
#define E(NAME, NI, WHAT) {#NAME, NAME, NI, WHAT}
static const Probe probes[] = {
  E(p_fma, 8, "v_fma_f32, 8 independent chains"),
  E(p_fma_dep, 8, "v_fma_f32, one dependent chain"),
  E(p_fma_sgpr, 8, "v_fma_f32 with an SGPR multiplicand"),
  E(p_add, 8, "v_add_f32"),
  E(p_add_dep, 8, "v_add_f32 dependent"),
  E(p_max, 8, "v_max_f32"),
  E(p_max3, 8, "v_max3_f32"),
  E(p_mov, 8, "v_mov_b32"),
  E(p_addu, 8, "v_add_u32"),
  E(p_lshl_or, 8, "v_lshl_or_b32"),
  E(p_pk_fma, 8, "v_pk_fma_f32 (VGPR operands)"),
  E(p_pk_fma_sgpr, 8, "v_pk_fma_f32, SGPR pair operand + op_sel_hi (as in the lean step)"),
  E(p_pk_add, 8, "v_pk_add_f32"),
  E(p_pk_mul, 8, "v_pk_mul_f32"),
  E(p_pk_add_dep, 8, "v_pk_add_f32 dependent"),
  E(p_sub_dpp_quad, 8, "v_sub_f32_dpp quad_perm:[1,0,3,2]"),
  E(p_max_dpp_quad, 8, "v_max_f32_dpp quad_perm:[1,0,3,2]"),
  E(p_max_dpp_ror8, 8, "v_max_f32_dpp row_ror:8"),
  E(p_max_dpp_hmir, 8, "v_max_f32_dpp row_half_mirror"),
  E(p_add_dpp_shr1, 8, "v_add_f32_dpp row_shr:1"),
  E(p_mov_dpp_quad, 8, "v_mov_b32_dpp quad_perm"),
  E(p_max_dpp_dep, 8, "s_nop 1 + v_max_f32_dpp dependent (counted as 8 = per pair)"),
  E(p_cmpclass_addc, 8, "v_cmp_class_f32 vcc -> v_addc_co_u32 (4 pairs = 8 instr)"),
  E(p_cmpgt_addc, 8, "v_cmp_gt_f32 vcc -> v_addc_co_u32 (4 pairs = 8 instr)"),
  E(p_cmpclass, 8, "v_cmp_class_f32_e32 vcc"),
  E(p_cmpgt, 8, "v_cmp_gt_f32_e32 vcc"),
  E(p_cmpgt_e64, 8, "v_cmp_gt_f32_e64 sgpr pair"),
  E(p_addc, 8, "v_addc_co_u32 (vcc in/out)"),
  E(p_cndmask, 8, "v_cndmask_b32 vcc"),
  E(p_swap16, 8, "v_permlane16_swap_b32"),
  E(p_swap32, 8, "v_permlane32_swap_b32"),
  E(p_swap16_dep, 8, "v_permlane16_swap_b32 dependent"),
  E(p_snop1_fma, 4, "s_nop 1 ; v_fma_f32 (4 pairs, counted as 4)"),
  E(p_smov_fma, 4, "s_mov_b32 ; v_fma_f32 (4 pairs, counted as 4)"),
  E(p_bpermute, 8, "ds_bpermute_b32 x8 + wait"),
  E(p_swizzle, 8, "ds_swizzle_b32 SWAP,16 x8 + wait"),
  E(p_step_cur, 1, "lean step q=0, 1 LLR: pk_fma, cmp_class, addc, sub_dpp, max_dpp  (cycles per STEP)"),
  E(p_step_2fma, 1, "same with two v_fma_f32 instead of v_pk_fma_f32             (cycles per STEP)"),
  E(p_step_swap, 1, "lean step q=4: pk_fma, cmp_gt, addc, permlane16_swap, max       (cycles per STEP)"),
  E(p_sub, 8, "v_sub_f32"), E(p_mul, 8, "v_mul_f32"), E(p_min, 8, "v_min_f32"),
  E(p_add_sgpr, 8, "v_add_f32 with an SGPR operand"),
  E(p_add_e64, 8, "v_add_f32_e64 (VOP3 encoding)"), E(p_add_neg, 8, "v_add_f32_e64 with a neg modifier"),
  E(p_fmac, 8, "v_fmac_f32"), E(p_fma_c0, 8, "v_fma_f32 a, b, 0"), E(p_fma_aab, 8, "v_fma_f32 a, b, b"),
  E(p_and, 8, "v_and_b32"), E(p_or, 8, "v_or_b32"), E(p_xor, 8, "v_xor_b32"),
  E(p_lshlrev, 8, "v_lshlrev_b32"), E(p_lshrrev, 8, "v_lshrrev_b32"), E(p_alignbit, 8, "v_alignbit_b32"),
  E(p_bfi, 8, "v_bfi_b32"), E(p_perm, 8, "v_perm_b32"),
  E(p_max_i32, 8, "v_max_i32"), E(p_max_u32, 8, "v_max_u32"), E(p_max_f16, 8, "v_max_f16"), E(p_pk_max_f16, 8, "v_pk_max_f16"),
  E(p_add_co, 8, "v_add_co_u32 (carry out to vcc)"),
  E(p_cmp_cndmask, 8, "v_cmp_gt_f32 vcc ; v_cndmask_b32 vcc (4 pairs = 8)"),
  E(p_cndmask_e64, 8, "v_cndmask_b32_e64 with an SGPR-pair mask"),
  E(p_cndmask_b, 8, "v_cndmask_b32 vcc, sources not rewritten"),
  E(p_max_noieee, 8, "v_max_f32 with MODE.IEEE = 0"),
  E(p_add_f64, 8, "v_add_f64"), E(p_fma_f64, 8, "v_fma_f64"), E(p_max_f64, 8, "v_max_f64"), E(p_mul_f64, 8, "v_mul_f64"),
  E(p_pk_mov, 8, "v_pk_mov_b32"),
  E(p_writelane, 8, "v_writelane_b32"), E(p_mov_sgpr, 8, "v_mov_b32 from an SGPR"),
  E(p_add_sdwa, 8, "v_add_f32_sdwa (dword selects)"),
  E(p_sub_dpp_b0, 8, "v_sub_f32_dpp / v_sub_f32 alternating"),
  E(p_ldsbc_fma, 8, "8 v_fma_f32 + one broadcast ds_read_b128 (counted as 8)"),
  E(p_step_sign, 1, "DPP step, sign trick: pk_fma, fma, alignbit, sub_dpp, max_dpp      (cycles per STEP)"),
  E(p_step_swap_sign, 1, "swap step, sign trick: pk_fma, sub, alignbit, swap, max         (cycles per STEP)"),
  E(p_step_int, 1, "DPP step, integer compare: pk_fma, addc, subb_dpp, max_u32_dpp            (cycles per STEP)"),
  E(p_step_int_cnd, 1, "DPP step, integer compare: pk_fma, addc, subb_dpp, cndmask_dpp        (cycles per STEP)"),
  E(p_step_swz, 1, "bit-4 step via ds_swizzle: pk_fma, swizzle, cmp_class, addc, sub, max      (cycles per STEP)"),
  E(p_step_swz_int, 1, "bit-4 step via ds_swizzle, integer: pk_fma, swizzle, addc, subb, max_u32 (cycles per STEP)"),
  E(p_step_bperm_int, 1, "bit-5 step via ds_bpermute, integer: pk_fma, bpermute, addc, subb, max_u32 (cycles per STEP)"),
  E(p_step_swap_int, 1, "swap step, integer: pk_fma, addc, swap16, sub_co, max_u32              (cycles per STEP)"),
  E(p_subb_dpp, 8, "v_subb_co_u32_dpp (vcc in/out)"),
  E(p_max_u32_dpp, 8, "v_max_u32_dpp quad_perm"),
  E(p_chain_dpp, 1, "metric chain only: fma, fma, s_nop 1, max_dpp                    (cycles per STEP)"),
};

int main(int argc, char** argv)
{
  const char* only = argc > 1 ? argv[1] : nullptr;
  const int iters = 2000;
  float* sink; Out* out; float* seed;
  const int maxw = 512 * 16;
  hipMalloc(&sink, (size_t)maxw * 64 * 4); hipMalloc(&out, maxw * sizeof(Out)); hipMalloc(&seed, 64);
  const float hs[4] = {1.0f, 0.5f, 1.0f, 0.0f};
  hipMemcpy(seed, hs, sizeof hs, hipMemcpyHostToDevice);
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("# CUs %d; iters %d x REP %d; cyc = shader cycles per wave-instruction per SIMD (median over waves | from wall clock)\n", ncu, iters, REP);
  printf("%-18s %-6s %22s %22s %22s %22s   %s\n", "probe", "", "W=1", "W=2", "W=4", "W=8", "what");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Probe& p : probes) {
    if (only && !strstr(p.name, only)) continue;
    hipFuncSetAttribute((const void*)p.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    printf("%-18s %-6s", p.name, "");
    for (int W : {1, 2, 4, 8}) {
      const int threads = W == 8 ? 1024 : 256 * W, blocks = W == 8 ? 2 * ncu : ncu;
      const size_t lds = W == 8 ? 72 * 1024 : 96 * 1024;      // one (W<8) or two (W=8) workgroups per CU, no more
      float best_ms = 1e9; double med = 0, clk = 0;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(threads), lds, 0, sink, out, seed, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * threads / 64;
        std::vector<Out> h(nw);
        hipMemcpy(h.data(), out, nw * sizeof(Out), hipMemcpyDeviceToHost);
        std::vector<double> c(nw); double sc = 0, sw = 0;
        for (int i = 0; i < nw; i++) { c[i] = (double)h[i].cyc; sc += h[i].cyc; sw += h[i].wall; }
        std::nth_element(c.begin(), c.begin() + nw / 2, c.end());
        if (ms < best_ms) { best_ms = ms; med = c[nw / 2]; clk = sc / sw * 100e6; }
      }
      const double ninstr = (double)iters * REP * p.ni;
      const double cyc_med = med / (W * ninstr);
      const double cyc_wall = best_ms * 1e-3 * clk / (W * ninstr);
      printf("  %6.2f | %6.2f @%4.2fGHz", cyc_med, cyc_wall, clk * 1e-9);
    }
    printf("   %s\n", p.what);
    fflush(stdout);
  }
  return 0;
}
