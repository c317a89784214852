// Issue cost of wave64 vector instructions on gfx950 with a full SIMD (8 waves each): cycles per instruction and SIMD.
// One kernel per instruction kind; every wave runs ITER x 64 instructions on eight independent accumulators and
// times itself with s_memtime; cost = elapsed / (ITER x 64 x waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o tools/probes/build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

#define BODY8(OP)                                                                         \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                          \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(x), "v"(y) : "vcc", "s10", "s11");

#define KERNEL(NAME, OP)                                                                  \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) { \
        uint32_t a[8];                                                                    \
        for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;                               \
        unsigned long long t0 = __builtin_readcyclecounter();                             \
        for (int it = 0; it < ITER; it++) {                                               \
            BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) BODY8(OP) \
        }                                                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                             \
        uint32_t s = 0;                                                                   \
        for (int i = 0; i < 8; i++) s ^= a[i];                                            \
        if (s == 0x12345678u) sink[0] = s;                                                \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

#define OP_0(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_0, OP_0)
#define OP_1(i) "v_fmac_f32 %" #i ", %8, %9\n"
KERNEL(k_1, OP_1)
#define OP_2(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_2, OP_2)
#define OP_3(i) "v_add_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_3, OP_3)
#define OP_4(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_4, OP_4)
#define OP_5(i) "v_min_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_5, OP_5)
#define OP_6(i) "v_max_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_6, OP_6)
#define OP_7(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_7, OP_7)
#define OP_8(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_8, OP_8)
#define OP_9(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_9, OP_9)
#define OP_10(i) "v_and_b32 %" #i ", %" #i ", %8\n"
KERNEL(k_10, OP_10)
#define OP_11(i) "v_or_b32 %" #i ", %" #i ", %8\n"
KERNEL(k_11, OP_11)
#define OP_12(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
KERNEL(k_12, OP_12)
#define OP_13(i) "v_not_b32 %" #i ", %" #i "\n"
KERNEL(k_13, OP_13)
#define OP_14(i) "v_add_u32 %" #i ", %" #i ", %8\n"
KERNEL(k_14, OP_14)
#define OP_15(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
KERNEL(k_15, OP_15)
#define OP_16(i) "v_mov_b32 %" #i ", %8\n"
KERNEL(k_16, OP_16)
#define OP_17(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
KERNEL(k_17, OP_17)
#define OP_18(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
KERNEL(k_18, OP_18)
#define OP_19(i) "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
KERNEL(k_19, OP_19)
#define OP_20(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_20, OP_20)
#define OP_21(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_21, OP_21)
#define OP_22(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
KERNEL(k_22, OP_22)
#define OP_23(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
KERNEL(k_23, OP_23)
#define OP_24(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_24, OP_24)
#define OP_25(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_25, OP_25)
#define OP_26(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_26, OP_26)
#define OP_27(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
KERNEL(k_27, OP_27)
#define OP_28(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
KERNEL(k_28, OP_28)
#define OP_29(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
KERNEL(k_29, OP_29)
#define OP_30(i) "v_ffbh_u32 %" #i ", %" #i "\n"
KERNEL(k_30, OP_30)
#define OP_31(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
KERNEL(k_31, OP_31)
#define OP_32(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
KERNEL(k_32, OP_32)
#define OP_33(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_33, OP_33)
#define OP_34(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
KERNEL(k_34, OP_34)
#define OP_35(i) "v_cvt_f32_ubyte3 %" #i ", %" #i "\n"
KERNEL(k_35, OP_35)
#define OP_36(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
KERNEL(k_36, OP_36)
#define OP_37(i) "v_cvt_u32_f32 %" #i ", %" #i "\n"
KERNEL(k_37, OP_37)
#define OP_38(i) "v_rcp_f32 %" #i ", %" #i "\n"
KERNEL(k_38, OP_38)
#define OP_39(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
KERNEL(k_39, OP_39)
#define OP_40(i) "v_cmp_lt_f32 s[10:11], %" #i ", %8\n"
KERNEL(k_40, OP_40)
#define OP_41(i) "v_cmp_eq_u32 vcc, %" #i ", %8\n"
KERNEL(k_41, OP_41)
#define OP_42(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
KERNEL(k_42, OP_42)
#define OP_43(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]\n"
KERNEL(k_43, OP_43)
#define OP_44(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
KERNEL(k_44, OP_44)
#define OP_45(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
KERNEL(k_45, OP_45)
#define OP_46(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
KERNEL(k_46, OP_46)
#define OP_47(i) "v_and_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
KERNEL(k_47, OP_47)
#define OP_48(i) "v_cvt_f32_u32_sdwa %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
KERNEL(k_48, OP_48)
#define OP_49(i) "v_max_u32 %" #i ", %" #i ", %8\n"
KERNEL(k_49, OP_49)
#define OP_50(i) "v_min_u32 %" #i ", %" #i ", %8\n"
KERNEL(k_50, OP_50)
#define OP_51(i) "v_max_i32 %" #i ", %" #i ", %8\n"
KERNEL(k_51, OP_51)
#define OP_52(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_52, OP_52)
#define OP_53(i) "v_cvt_pk_u8_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_53, OP_53)
#define OP_54(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_54, OP_54)
#define OP_55(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_55, OP_55)
#define OP_56(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_56, OP_56)
#define OP_57(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_57, OP_57)
#define OP_58(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_58, OP_58)
#define OP_59(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
KERNEL(k_59, OP_59)
#define OP_60(i) "v_pk_max_u16 %" #i ", %" #i ", %8\n"
KERNEL(k_60, OP_60)
#define OP_61(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
KERNEL(k_61, OP_61)
#define OP_62(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
KERNEL(k_62, OP_62)
#define OP_63(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_63, OP_63)
#define OP_64(i) "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
KERNEL(k_64, OP_64)
#define OP_65(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_65, OP_65)
#define OP_66(i) "v_cmp_class_f32 vcc, %" #i ", %8\n"
KERNEL(k_66, OP_66)
#define OP_67(i) "v_fract_f32 %" #i ", %" #i "\n"
KERNEL(k_67, OP_67)
#define OP_68(i) "v_ldexp_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_68, OP_68)
#define OP_69(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_69, OP_69)

#define OP_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
KERNEL(k_ashr, OP_ASHR)
#define OP_BFM(i) "v_bfm_b32 %" #i ", %" #i ", %8\n"
KERNEL(k_bfm, OP_BFM)
#define OP_SUBREV(i) "v_subrev_u32 %" #i ", 1, %" #i "\n"
KERNEL(k_subrev, OP_SUBREV)

// packed f32 forms work on register pairs
#define BODY8W(OP)                                                                        \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                          \
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) \
                 : "v"(xx), "v"(yy));
#define KERNELW(NAME, OP)                                                                 \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) { \
        unsigned long long b[8], xx = x | ((unsigned long long)y << 32), yy = y | ((unsigned long long)x << 32); \
        for (int i = 0; i < 8; i++) b[i] = threadIdx.x + i;                               \
        unsigned long long t0 = __builtin_readcyclecounter();                             \
        for (int it = 0; it < ITER; it++) {                                               \
            BODY8W(OP) BODY8W(OP) BODY8W(OP) BODY8W(OP) BODY8W(OP) BODY8W(OP) BODY8W(OP) BODY8W(OP) \
        }                                                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                             \
        unsigned long long s = 0;                                                         \
        for (int i = 0; i < 8; i++) s ^= b[i];                                            \
        if (s == 0x12345678ull) sink[0] = (uint32_t)s;                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }
#define OP_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_PKFMA32B(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9 op_sel_hi:[1,0,0]\n"
#define OP_PKMUL32(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define OP_PKADD32(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define OP_PKMOV(i) "v_pk_mov_b32 %" #i ", %" #i ", %8\n"
#define OP_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define OP_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define OP_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %8\n"
KERNELW(k_pkfma32, OP_PKFMA32)
KERNELW(k_pkfma32b, OP_PKFMA32B)
KERNELW(k_pkmul32, OP_PKMUL32)
KERNELW(k_pkadd32, OP_PKADD32)
KERNELW(k_pkmov, OP_PKMOV)
KERNELW(k_fma64, OP_FMA64)
KERNELW(k_add64, OP_ADD64)
KERNELW(k_mul64, OP_MUL64)
KERNELW(k_lshladd64, OP_LSHLADD64)

// mixes of a 2-cycle and a 4-cycle kind: do the two kinds share one issue port (6 cycles per pair) or not (4)?
#define OP_MIX1(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\nv_min_f32 %" #i ", %" #i ", %8\n"
#define OP_MIX2(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\nv_fma_f32 %" #i ", %" #i ", %9, %8\nv_min_f32 %" #i ", %" #i ", %8\n"
#define OP_MIX3(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\nv_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MIX4(i) "v_and_b32 %" #i ", %" #i ", %8\nv_lshlrev_b32 %" #i ", 1, %" #i "\n"
KERNEL(k_mix1, OP_MIX1)
KERNEL(k_mix2, OP_MIX2)
KERNEL(k_mix3, OP_MIX3)
KERNEL(k_mix4, OP_MIX4)

#define OPX_0(i) "v_fma_f32 %0, %0, %8, %9\nv_min_f32 %4, %4, %8\nv_fma_f32 %1, %1, %8, %9\nv_min_f32 %5, %5, %8\nv_fma_f32 %2, %2, %8, %9\nv_min_f32 %6, %6, %8\nv_fma_f32 %3, %3, %8, %9\nv_min_f32 %7, %7, %8\nv_fma_f32 %4, %4, %8, %9\nv_min_f32 %0, %0, %8\nv_fma_f32 %5, %5, %8, %9\nv_min_f32 %1, %1, %8\nv_fma_f32 %6, %6, %8, %9\nv_min_f32 %2, %2, %8\nv_fma_f32 %7, %7, %8, %9\nv_min_f32 %3, %3, %8\n"
#define BODYX_0 asm volatile(OPX_0(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_0(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_0 BODYX_0 BODYX_0 BODYX_0 BODYX_0 BODYX_0 BODYX_0 BODYX_0 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
#define OPX_1(i) "v_and_b32 %0, %0, %8\nv_min_f32 %4, %4, %8\nv_and_b32 %1, %1, %8\nv_min_f32 %5, %5, %8\nv_and_b32 %2, %2, %8\nv_min_f32 %6, %6, %8\nv_and_b32 %3, %3, %8\nv_min_f32 %7, %7, %8\nv_and_b32 %4, %4, %8\nv_min_f32 %0, %0, %8\nv_and_b32 %5, %5, %8\nv_min_f32 %1, %1, %8\nv_and_b32 %6, %6, %8\nv_min_f32 %2, %2, %8\nv_and_b32 %7, %7, %8\nv_min_f32 %3, %3, %8\n"
#define BODYX_1 asm volatile(OPX_1(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_1(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_1 BODYX_1 BODYX_1 BODYX_1 BODYX_1 BODYX_1 BODYX_1 BODYX_1 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
#define OPX_2(i) "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %4, %4\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %5, %5\nv_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %6, %6\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %7, %7\nv_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %0, %0\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %1, %1\nv_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %2, %2\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\nv_cvt_f32_ubyte1 %3, %3\n"
#define BODYX_2 asm volatile(OPX_2(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_2(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_2 BODYX_2 BODYX_2 BODYX_2 BODYX_2 BODYX_2 BODYX_2 BODYX_2 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
#define OPX_3(i) "v_mov_b32 %0, %8\nv_lshlrev_b32 %4, 1, %4\nv_mov_b32 %1, %8\nv_lshlrev_b32 %5, 1, %5\nv_mov_b32 %2, %8\nv_lshlrev_b32 %6, 1, %6\nv_mov_b32 %3, %8\nv_lshlrev_b32 %7, 1, %7\nv_mov_b32 %4, %8\nv_lshlrev_b32 %0, 1, %0\nv_mov_b32 %5, %8\nv_lshlrev_b32 %1, 1, %1\nv_mov_b32 %6, %8\nv_lshlrev_b32 %2, 1, %2\nv_mov_b32 %7, %8\nv_lshlrev_b32 %3, 1, %3\n"
#define BODYX_3 asm volatile(OPX_3(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_3(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_3 BODYX_3 BODYX_3 BODYX_3 BODYX_3 BODYX_3 BODYX_3 BODYX_3 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
#define OPX_4(i) "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %4, %4, %9, %8\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %5, %5, %9, %8\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %6, %6, %9, %8\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %7, %7, %9, %8\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %0, %0, %9, %8\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %1, %1, %9, %8\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %2, %2, %9, %8\nv_fma_f32 %7, %7, %8, %9\nv_fma_f32 %3, %3, %9, %8\n"
#define BODYX_4 asm volatile(OPX_4(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_4(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_4 BODYX_4 BODYX_4 BODYX_4 BODYX_4 BODYX_4 BODYX_4 BODYX_4 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
#define OPX_5(i) "v_min_f32 %0, %0, %8\nv_cvt_f32_ubyte1 %4, %4\nv_min_f32 %1, %1, %8\nv_cvt_f32_ubyte1 %5, %5\nv_min_f32 %2, %2, %8\nv_cvt_f32_ubyte1 %6, %6\nv_min_f32 %3, %3, %8\nv_cvt_f32_ubyte1 %7, %7\nv_min_f32 %4, %4, %8\nv_cvt_f32_ubyte1 %0, %0\nv_min_f32 %5, %5, %8\nv_cvt_f32_ubyte1 %1, %1\nv_min_f32 %6, %6, %8\nv_cvt_f32_ubyte1 %2, %2\nv_min_f32 %7, %7, %8\nv_cvt_f32_ubyte1 %3, %3\n"
#define BODYX_5 asm volatile(OPX_5(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_5(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_5 BODYX_5 BODYX_5 BODYX_5 BODYX_5 BODYX_5 BODYX_5 BODYX_5 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
#define OPX_6(i) "v_add_u32 %0, %0, %8\nv_cmp_lt_f32 vcc, %4, %8\nv_add_u32 %1, %1, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_add_u32 %2, %2, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_add_u32 %3, %3, %8\nv_cmp_lt_f32 vcc, %7, %8\nv_add_u32 %4, %4, %8\nv_cmp_lt_f32 vcc, %0, %8\nv_add_u32 %5, %5, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_add_u32 %6, %6, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_add_u32 %7, %7, %8\nv_cmp_lt_f32 vcc, %3, %8\n"
#define BODYX_6 asm volatile(OPX_6(0) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y) : "vcc");
__global__ __launch_bounds__(256) void kx_6(unsigned long long* out, uint32_t* sink, uint32_t x, uint32_t y) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) { BODYX_6 BODYX_6 BODYX_6 BODYX_6 BODYX_6 BODYX_6 BODYX_6 BODYX_6 }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, uint32_t*, uint32_t, uint32_t);

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    const int cus = p.multiProcessorCount;
    unsigned long long* d_out;
    uint32_t* d_sink;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_fma_f32", k_0},
        {"v_fmac_f32", k_1},
        {"v_mul_f32", k_2},
        {"v_add_f32", k_3},
        {"v_sub_f32", k_4},
        {"v_min_f32", k_5},
        {"v_max_f32", k_6},
        {"v_min3_f32", k_7},
        {"v_max3_f32", k_8},
        {"v_med3_f32", k_9},
        {"v_and_b32", k_10},
        {"v_or_b32", k_11},
        {"v_xor_b32", k_12},
        {"v_not_b32", k_13},
        {"v_add_u32", k_14},
        {"v_sub_u32", k_15},
        {"v_mov_b32", k_16},
        {"v_lshlrev_b32", k_17},
        {"v_lshrrev_b32", k_18},
        {"v_bfe_u32", k_19},
        {"v_bfi_b32", k_20},
        {"v_perm_b32", k_21},
        {"v_lshl_or_b32", k_22},
        {"v_lshl_add_u32", k_23},
        {"v_and_or_b32", k_24},
        {"v_or3_b32", k_25},
        {"v_add3_u32", k_26},
        {"v_bitop3_b32", k_27},
        {"v_bcnt_u32_b32", k_28},
        {"v_mbcnt_lo_u32_b32", k_29},
        {"v_ffbh_u32", k_30},
        {"v_mul_lo_u32", k_31},
        {"v_mul_u32_u24", k_32},
        {"v_mad_u32_u24", k_33},
        {"v_cvt_f32_ubyte0", k_34},
        {"v_cvt_f32_ubyte3", k_35},
        {"v_cvt_f32_u32", k_36},
        {"v_cvt_u32_f32", k_37},
        {"v_rcp_f32", k_38},
        {"v_cmp_lt_f32_vcc", k_39},
        {"v_cmp_lt_f32_sgpr", k_40},
        {"v_cmp_eq_u32_vcc", k_41},
        {"v_cndmask_vcc", k_42},
        {"v_cndmask_sgpr", k_43},
        {"cmp+cndmask", k_44},
        {"v_mov_b32_dpp", k_45},
        {"v_add_f32_dpp", k_46},
        {"v_and_b32_sdwa", k_47},
        {"v_cvt_f32_ubyte_sdwa", k_48},
        {"v_max_u32", k_49},
        {"v_min_u32", k_50},
        {"v_max_i32", k_51},
        {"v_sad_u8", k_52},
        {"v_cvt_pk_u8_f32", k_53},
        {"v_cvt_pkrtz_f16_f32", k_54},
        {"v_pk_fma_f16", k_55},
        {"v_pk_mul_f16", k_56},
        {"v_pk_min_f16", k_57},
        {"v_pk_max_f16", k_58},
        {"v_pk_add_u16", k_59},
        {"v_pk_max_u16", k_60},
        {"v_pk_min_u16", k_61},
        {"v_pk_mul_lo_u16", k_62},
        {"v_pk_mad_u16", k_63},
        {"v_pk_lshlrev_b16", k_64},
        {"v_dot4_u32_u8", k_65},
        {"v_cmp_class_f32", k_66},
        {"v_fract_f32", k_67},
        {"v_ldexp_f32", k_68},
        {"v_mad_i32_i24", k_69},
        {"v_ashrrev_i32", k_ashr}, {"v_bfm_b32", k_bfm}, {"v_subrev_u32", k_subrev},
        {"v_pk_fma_f32", k_pkfma32}, {"v_pk_fma_f32 op_sel", k_pkfma32b}, {"v_pk_mul_f32", k_pkmul32}, {"v_pk_add_f32", k_pkadd32},
        {"v_pk_mov_b32", k_pkmov}, {"v_fma_f64", k_fma64}, {"v_add_f64", k_add64}, {"v_mul_f64", k_mul64}, {"v_lshl_add_u64", k_lshladd64},
        {"fma+min (per pair)", k_mix1}, {"fma+fma+min (per triple)", k_mix2}, {"cvt_ubyte+fma (per pair)", k_mix3}, {"and+lshl (per pair)", k_mix4}, {"indep: fma+min (per pair)", kx_0}, {"indep: and+min (per pair)", kx_1}, {"indep: bitop3+cvt (per pair)", kx_2}, {"indep: mov+lshl (per pair)", kx_3}, {"indep: fma+fma (per pair)", kx_4}, {"indep: min+cvt (per pair)", kx_5}, {"indep: add_u32+cmp (per pair)", kx_6}};
    for (int wps : {8, 1}) {  // waves per SIMD
        const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
        const int waves = blocks * 4;
        hipMalloc(&d_out, waves * sizeof(unsigned long long));
        hipMalloc(&d_sink, 4);
        std::vector<unsigned long long> h(waves);
        for (auto& e : ks) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 3u, 5u);  // warm-up
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, d_sink, 3u, 5u);
            hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += (double)v;
            mean /= waves;
            const double n = (double)ITER * 64.0;
            // s_memtime ticks at a constant 100 MHz on this chip; the launch time gives the cost in ns
            printf("%d waves/SIMD  %-18s ticks/instr/wave %.3f   ns per instr and SIMD %.4f\n", wps, e.name, mean / n,
                   (double)ms * 1e6 / (n * wps));
        }
        hipFree(d_out);
        hipFree(d_sink);
    }
    return 0;
}
