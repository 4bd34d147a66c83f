#!/usr/bin/env python3
"""Generates valu_rate.hip: one kernel per candidate instruction (16 independent chains, unrolled), timed with
s_memtime inside the wave.  Prints issue cost in cycles per wave-instruction per SIMD at 1/2/4/8 waves per SIMD.
Diagnostic only (tools/diag); used to size the FAST kernel of round 2."""
import sys
OPS = [
    ("v_and_b32",        "v_and_b32 {d}, {a}, {d}"),
    ("v_add_u32",        "v_add_u32 {d}, {a}, {d}"),
    ("v_fma_f32",        "v_fma_f32 {d}, {a}, {b}, {d}"),
    ("v_lerp_u8",        "v_lerp_u8 {d}, {a}, {d}, {b}"),
    ("v_perm_b32",       "v_perm_b32 {d}, {a}, {d}, {b}"),
    ("v_alignbyte_b32",  "v_alignbyte_b32 {d}, {a}, {d}, 1"),
    ("v_alignbit_b32",   "v_alignbit_b32 {d}, {a}, {d}, 16"),
    ("v_bitop3_b32",     "v_bitop3_b32 {d}, {a}, {d}, {b} bitop3:0x96"),
    ("v_and_or_b32",     "v_and_or_b32 {d}, {a}, {d}, {b}"),
    ("v_or3_b32",        "v_or3_b32 {d}, {a}, {d}, {b}"),
    ("v_bfi_b32",        "v_bfi_b32 {d}, {a}, {d}, {b}"),
    ("v_max3_u32",       "v_max3_u32 {d}, {a}, {d}, {b}"),
    ("v_min3_i32",       "v_min3_i32 {d}, {a}, {d}, {b}"),
    ("v_max3_u16",       "v_max3_u16 {d}, {a}, {d}, {b}"),
    ("v_pk_max_u16",     "v_pk_max_u16 {d}, {a}, {d}"),
    ("v_pk_min_i16",     "v_pk_min_i16 {d}, {a}, {d}"),
    ("v_pk_sub_u16_clamp", "v_pk_sub_u16 {d}, {a}, {d} clamp"),
    ("v_pk_max_f16",     "v_pk_max_f16 {d}, {a}, {d}"),
    ("v_pk_maximum3_f16", "v_pk_maximum3_f16 {d}, {a}, {d}, {b}"),
    ("v_pk_minimum3_f16", "v_pk_minimum3_f16 {d}, {a}, {d}, {b}"),
    ("v_pk_add_f16",     "v_pk_add_f16 {d}, {a}, {d}"),
    ("v_pk_fma_f32",     "v_pk_fma_f32 {d2}, {a2}, {b2}, {d2}"),
    ("v_sad_u8",         "v_sad_u8 {d}, {a}, {d}, {b}"),
    ("v_dot4_u32_u8",    "v_dot4_u32_u8 {d}, {a}, {d}, {b}"),
    ("v_mov_dpp_row_shr", "v_mov_b32_dpp {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_mov_dpp_wave_shr", "v_mov_b32_dpp {d}, {d} wave_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_add_dpp_wave_shr", "v_add_u32_dpp {d}, {d}, {a} wave_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_cmp_gt_u32(vcc)", "v_cmp_gt_u32 vcc, {a}, {d}"),
    ("v_cmp_sdwa_byte",  "v_cmp_gt_u32_sdwa vcc, {a}, {d} src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_cndmask_b32",    "v_cndmask_b32 {d}, {a}, {d}, vcc"),
    ("v_mbcnt_lo",       "v_mbcnt_lo_u32_b32 {d}, s20, {d}"),
    ("v_mul_lo_u32",     "v_mul_lo_u32 {d}, {a}, {d}"),
    ("v_mul_u32_u24",    "v_mul_u32_u24 {d}, {a}, {d}"),
    ("v_mad_u32_u24",    "v_mad_u32_u24 {d}, {a}, {d}, {b}"),
    ("v_lshl_or_b32",    "v_lshl_or_b32 {d}, {a}, 8, {d}"),
    ("v_bfe_u32",        "v_bfe_u32 {d}, {d}, 8, 8"),
    ("v_readlane",       "v_readlane_b32 s21, {d}, 3"),
    ("s_and_b32(salu)",  "s_and_b32 s21, s21, s22"),
    ("s_bcnt1_i32_b64",  "s_bcnt1_i32_b64 s21, s[22:23]"),
    ("ds_read_u8",       "ds_read_u8 {d}, {addr}"),
    ("ds_read_u8_d16_hi", "ds_read_u8_d16_hi {d}, {addr}"),
    ("ds_read_b32",      "ds_read_b32 {d}, {addr}"),
    ("ds_read_b64",      "ds_read_b64 {d2}, {addr}"),
    ("ds_read_b128",     "ds_read_b128 {d4}, {addr}"),
    ("ds_write_b32",     "ds_write_b32 {addr}, {a}"),
    ("ds_write_b16",     "ds_write_b16 {addr}, {a}"),
    ("ds_write_b8",      "ds_write_b8 {addr}, {a}"),
    ("v_or_b32",         "v_or_b32 {d}, {a}, {d}"),
    ("v_xor_b32",        "v_xor_b32 {d}, {a}, {d}"),
    ("v_not_b32",        "v_not_b32 {d}, {d}"),
    ("v_mov_b32",        "v_mov_b32 {d}, {a}"),
    ("v_lshlrev_b32",    "v_lshlrev_b32 {d}, 3, {d}"),
    ("v_lshrrev_b32",    "v_lshrrev_b32 {d}, 3, {d}"),
    ("v_sub_u32",        "v_sub_u32 {d}, {a}, {d}"),
    ("v_min_u32",        "v_min_u32 {d}, {a}, {d}"),
    ("v_max_i32",        "v_max_i32 {d}, {a}, {d}"),
    ("v_min_u16",        "v_min_u16 {d}, {a}, {d}"),
    ("v_max_u32_sdwa",   "v_max_u32_sdwa {d}, {a}, {d} dst_sel:DWORD src0_sel:BYTE_1 src1_sel:BYTE_2"),
    ("v_and_b32_sdwa",   "v_and_b32_sdwa {d}, {a}, {d} dst_sel:DWORD src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_lshl_add_u32",   "v_lshl_add_u32 {d}, {d}, 1, {a}"),
    ("v_add3_u32",       "v_add3_u32 {d}, {d}, {a}, {b}"),
    ("v_xad_u32",        "v_xad_u32 {d}, {d}, {a}, {b}"),
    ("v_bcnt_u32_b32",   "v_bcnt_u32_b32 {d}, {d}, {a}"),
    ("v_ffbl_b32",       "v_ffbl_b32 {d}, {d}"),
    ("v_cndmask_sgpr",   "v_cndmask_b32 {d}, {a}, {d}, s[22:23]"),
    ("v_cmp_ne_sgpr",    "v_cmp_ne_u32 s[24:25], {a}, {d}"),
    ("v_mbcnt_hi",       "v_mbcnt_hi_u32_b32 {d}, s20, {d}"),
    ("ds_read_u8_scatter", "ds_read_u8 {d}, v9"),
    ("ds_read_u8_scat260", "ds_read_u8 {d}, v10"),
    ("ds_read2_b32",     "ds_read2_b32 {d2}, {addr} offset1:2"),
    ("ds_write_b8_scatter", "ds_write_b8 v9, {a}"),
    # mixes: does another instruction class issue beside VALU from the SAME wave / other waves?
    ("mix_valu+salu",    "v_and_b32 {d}, {a}, {d}\n s_and_b32 s21, s21, s22"),
    ("mix_valu+ds_read", "v_and_b32 {d}, {a}, {d}\n ds_read_u8 {e}, {addr}"),
    ("mix_2valu+ds_read", "v_and_b32 {d}, {a}, {d}\n v_lerp_u8 {d}, {a}, {d}, {b}\n ds_read_u8 {e}, {addr}"),
    ("mix_3valu+ds_read", "v_and_b32 {d}, {a}, {d}\n v_lerp_u8 {d}, {a}, {d}, {b}\n v_perm_b32 {d}, {a}, {d}, {b}\n ds_read_u8 {e}, {addr}"),
    ("mix_pk3+ds_read", "v_pk_maximum3_f16 {d}, {a}, {d}, {b}\n v_pk_minimum3_f16 {d}, {a}, {d}, {b}\n ds_read_u8_d16_hi {e}, {addr}"),
]
NCH = 16      # independent chains
ITER = 256
src = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <vector>', '#include <string>', '#include <algorithm>',
       '#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)']
names = []
for idx, (name, tmpl) in enumerate(OPS):
    names.append(name)
    ninst = len(tmpl.split("\n"))
    body = []
    for i in range(NCH):
        # chains use v[32+2i .. ] for d (pairs / quads when needed), v[4..7] as constant sources, v[8+i] as ds dest for mixes
        d = f"v{40 + 4 * i}"
        s = tmpl.format(d=d, d2=f"v[{40 + 4 * i}:{41 + 4 * i}]", d4=f"v[{40 + 4 * i}:{43 + 4 * i}]", a="v4", b="v5", c="v6",
                        a2="v[4:5]", b2="v[6:7]", addr="v8", e=f"v{12 + i}")
        body.append(s)
    asm_body = "\\n\\t".join("\\n\\t".join(b.split("\n")) for b in body)
    clob = ", ".join(f'"v{r}"' for r in list(range(4, 11)) + list(range(12, 28)) + list(range(40, 40 + 4 * NCH))) + ', "s20", "s21", "s22", "s23", "s24", "s25", "vcc", "memory"'
    src.append(f'''
__global__ __launch_bounds__(256) void k{idx}(unsigned long long* out, int iters) {{
    __shared__ unsigned int lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    asm volatile("v_mov_b32 v4, 0x01020304\\n\\tv_mov_b32 v5, 0x00010203\\n\\tv_mov_b32 v6, 0x3c003c00\\n\\tv_mov_b32 v7, 0\\n\\t"
                 "v_and_b32 v8, 63, %0\\n\\tv_mul_u32_u24 v9, 0x9E3779, v8\\n\\tv_lshrrev_b32 v9, 7, v9\\n\\tv_and_b32 v9, 0xfff, v9\\n\\tv_mul_u32_u24 v10, 0x5bd1e9, v8\\n\\tv_lshrrev_b32 v10, 9, v10\\n\\tv_and_b32 v10, 15, v10\\n\\tv_mul_u32_u24 v10, 260, v10\\n\\tv_lshl_add_u32 v10, v8, 2, v10\\n\\tv_lshlrev_b32 v8, 4, v8\\n\\ts_mov_b32 s20, -1\\n\\ts_mov_b32 s21, 5\\n\\ts_mov_b32 s22, 7\\n\\ts_mov_b32 s23, 9"
                 :: "v"(threadIdx.x) : {clob});
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {{
        asm volatile("{asm_body}" ::: {clob});
    }}
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}}''')
src.append('typedef void (*kfn)(unsigned long long*, int);')
src.append('static kfn KS[] = {' + ", ".join(f"k{i}" for i in range(len(OPS))) + '};')
src.append('static const char* NAMES[] = {' + ", ".join(f'"{n}"' for n in names) + '};')
src.append('static const int NINST[] = {' + ", ".join(str(len(t.split(chr(10)))) for _, t in OPS) + '};')
src.append(f'''
int main() {{
    const int NCH = {NCH}, ITER = {ITER};
    unsigned long long* d; CK(hipMalloc(&d, 8 * 4 * 256 * 8));
    std::vector<unsigned long long> h(4 * 256 * 8);
    printf("%-22s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD; 1/2/4/8 waves per SIMD)\\n", "op", "w1", "w2", "w4", "w8");
    for (size_t k = 0; k < sizeof(KS) / sizeof(KS[0]); ++k) {{
        printf("%-22s", NAMES[k]);
        double wall4 = 0;
        for (int w = 1; w <= 8; w *= 2) {{
            const int blocks = 256 * w;   // 256-thread blocks: one wave per SIMD each
            hipLaunchKernelGGL(KS[k], dim3(blocks), dim3(256), 0, 0, d, 8);   // warm
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(KS[k], dim3(blocks), dim3(256), 0, 0, d, ITER * 8);
            hipEventRecord(e1, 0);
            CK(hipDeviceSynchronize());
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (w == 4) wall4 = (double)ms * 1e6 / ((double)ITER * 8 * NCH * NINST[k] * w);   // ns per wave-instruction per SIMD
            hipLaunchKernelGGL(KS[k], dim3(blocks), dim3(256), 0, 0, d, ITER);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d, 8 * 4 * blocks, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + 4 * blocks);
            const double med = (double)h[2 * blocks];
            // every wave issues ITER * NCH * NINST instructions in `med` cycles beside w - 1 others on its SIMD
            printf(" %8.2f", med / ((double)ITER * NCH * NINST[k] * w));
        }}
        printf("   wall@w4 %.3f ns/instr/SIMD\\n", wall4);
        fflush(stdout);
    }}
    return 0;
}}''')
open(sys.argv[1] if len(sys.argv) > 1 else "valu_rate.hip", "w").write("\n".join(src))
