// fp8 attention forward, one wave per SIMD, on v_mfma_scale_f32_32x32x64_f8f6f4.
//
// Same arithmetic, operand maps and K/V tile image as the 8-wave kernel in attn_fp8.hip (which also
// holds the quantisation pre-pass); what changes is who fills the MFMA's shadow.  There two waves
// per SIMD alternate between an MFMA phase and a softmax phase and the hardware interleaves them
// as it can; here a lone wave issues, in program order, the MFMAs of one tile and, in the gap
// behind each of them, a fixed share of the softmax of the previous tile:
//
//   workgroup = 4 waves x 64 query rows; a wave holds two blocks (qb) of 32 rows, so every K and V^T
//   fragment it reads from LDS feeds two MFMAs.
//   period t:  QK(t+1): S(t+1) = K(t+1) (cQ)^T - m      4 fragments (kb, ks) x 2 qb = 8 MFMAs
//              SM(t):   P(t) = e4m3(exp2(S(t))), lane row sums              64 exp, 64 add, 32 convert
//              PV(t-1): O^T += V(t-1)^T P(t-1)^T         4 fragments (db) x 2 qb = 8 MFMAs
//   K and V^T fragments alternate (K00 V0 K10 V1 K01 V2 K11 V3), so no MFMA waits for the one issued
//   before it; gap G of the 16 carries the four exps of P dword G and the adds and conversions of
//   dword G - 1 (consumers run one gap behind their transcendentals).
//
// Fragments are prefetched three steps ahead into a ring of four 8-register slots.  The ring is a
// fixed block of accumulator registers, a[224:255], that only the asm statements below name: an
// LDS read into compiler-visible values would have to be waited for where the compiler sees fit,
// and it copies such values through VGPRs.  Completion is counted by hand (s_waitcnt lgkmcnt).
//
// K and V^T parts of the tile image live in separate LDS rings of 6 slots: K(t) is read during
// period t - 1 only and V(t) during period t + 1 only, so a slot is refilled (LDS-DMA, counted by
// hand with vmcnt) right after the tile barrier that follows its last read, five periods before
// the data is needed.
//
// Running maximum as accumulator init and the overflow check instead of a tracked maximum are
// those of the other attention kernels: a lane whose 32 exps of a row sum to at most 2^6 holds no P
// above that, well inside e4m3; otherwise (and on tile 0) the rare path moves everything to the
// row maximum of the tile and redoes P.
#include "attn_fp8_common.h"
#include "attn_w4_common.h"

using namespace lx2v_attn;
using namespace lx2v_attn_fp8;

namespace {

#ifndef F8W4_ABL
#define F8W4_ABL 0  // timing ablations (results are wrong with any bit set): 1 softmax VALU, 2 fragment reads, 4 staging, 8 barrier, 16 overflow check, 32 no waits for fragments, 64 no scale pieces in the staging, 128 exps only, 256 no exps
#endif

constexpr int F8_THREADS = 256;
constexpr int NSLOT = 6;
constexpr int K_RING = 0;
constexpr int V_RING = NSLOT * K_PART;               // 52224
constexpr int LDS_BYTES = NSLOT * (K_PART + V_PART);  // 102912
constexpr int DMA_PER_PERIOD = (F8W4_ABL & 64) ? 4 : 6;                    // per wave: 2 K rows, 1 K scales, 2 V^T rows, 1 V scales
constexpr float kMaxLaneSum = 64.0f;

// The fragment ring lives in a[224:255] by name: data flows from the reads that fill a slot to the
// MFMAs that consume it without the compiler seeing a value, so between the two it takes the
// registers for free and may park a spilled value there.  Every statement the gaps are made of
// names the whole ring as clobbered: nothing that lives across a gap can then be placed in it.
// a[216:218] belong to it as well: under LIN the scale dwords of the next tiles land there.
#define F8_RING_CLOBBERS "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
// ---- the fragment ring: a[224 + 8 slot + 4 half ..] ----
template <int SLOT, int HALF, int OFF>
LX2V_DEVICE void ring_read(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
#define F8_RD(S, H, LO, HI, C0, C1, C2, C3) \
  if constexpr (SLOT == S && HALF == H) asm volatile("ds_read_b128 a[" #LO ":" #HI "], %0 offset:%1" ::"v"(addr), "i"(OFF) : C0, C1, C2, C3)
  F8_RD(0, 0, 224, 227, "a224", "a225", "a226", "a227");
  F8_RD(0, 1, 228, 231, "a228", "a229", "a230", "a231");
  F8_RD(1, 0, 232, 235, "a232", "a233", "a234", "a235");
  F8_RD(1, 1, 236, 239, "a236", "a237", "a238", "a239");
  F8_RD(2, 0, 240, 243, "a240", "a241", "a242", "a243");
  F8_RD(2, 1, 244, 247, "a244", "a245", "a246", "a247");
  F8_RD(3, 0, 248, 251, "a248", "a249", "a250", "a251");
  F8_RD(3, 1, 252, 255, "a252", "a253", "a254", "a255");
#undef F8_RD
}

// S (VGPR) = K fragment (ring) x Q^T (AGPR) + (-m) (VGPR) for the first 64 channels, += for the others.
// TIED: the first step names S as read and written although it only writes it: a score tuple is then one
// value from the prologue to the end and keeps its registers, instead of a new value per tile that the
// compiler may place elsewhere and move there -- with a copy it could put right behind an MFMA
// whose result it believes to be there already.
// Scale bytes: byte ks of the K scale dword and of the Q scale dword.
template <int SLOT, int KS, bool TIED>
LX2V_DEVICE void mfma_qk(f32x16& s, const i32x8& qf, const f32x16& negm, int ksc, int qsc) {
#define F8_QK(S, REGS)                                                                                                                  \
  if constexpr (SLOT == S && KS == 0 && TIED)                                                                                           \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, " REGS ", %1, %2, %3, %4 op_sel_hi:[0,0,0]" : "+v"(s) : "a"(qf), "v"(negm), "v"(ksc), "v"(qsc)); \
  if constexpr (SLOT == S && KS == 0 && !TIED)                                                                                          \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, " REGS ", %1, %2, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(s) : "a"(qf), "v"(negm), "v"(ksc), "v"(qsc)); \
  if constexpr (SLOT == S && KS == 1)                                                                                                   \
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, " REGS ", %1, %0, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,0,0]" : "+v"(s) : "a"(qf), "v"(ksc), "v"(qsc))
  F8_QK(0, "a[224:231]");
  F8_QK(1, "a[232:239]");
  F8_QK(2, "a[240:247]");
  F8_QK(3, "a[248:255]");
#undef F8_QK
}

// O^T (AGPR) += V^T fragment (ring) x P^T (VGPR); scale byte db of the V scale dword, unit scale for P.
template <int SLOT, int DB>
LX2V_DEVICE void mfma_pv(f32x16& o, const i32x8& pf, int vsc, int unit) {
#define F8_PV1(S, D, REGS, SEL) \
  if constexpr (SLOT == S && DB == D) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, " REGS ", %1, %0, %2, %3 " SEL : "+a"(o) : "v"(pf), "v"(vsc), "v"(unit))
#define F8_PV(S, REGS)                                         \
  F8_PV1(S, 0, REGS, "op_sel_hi:[0,0,0]");                     \
  F8_PV1(S, 1, REGS, "op_sel:[1,0,0] op_sel_hi:[0,0,0]");      \
  F8_PV1(S, 2, REGS, "op_sel_hi:[1,0,0]");                     \
  F8_PV1(S, 3, REGS, "op_sel:[1,0,0] op_sel_hi:[1,0,0]")
  F8_PV(0, "a[224:231]");
  F8_PV(1, "a[232:239]");
  F8_PV(2, "a[240:247]");
  F8_PV(3, "a[248:255]");
#undef F8_PV
#undef F8_PV1
}

template <int N>
LX2V_DEVICE void wait_lds() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
}

// A scale dword read that the compiler gets no value from: it lands in a[216 + SLOT] (K block 0, K
// block 1, V) and is fetched by scale_fetch behind the wait that covers it.  (As a "=v" output the
// compiler holds a value from the moment the read is ISSUED, and under register pressure parks it in
// another register at once -- taking what the register held before the data landed.)
template <int SLOT, int OFF>
LX2V_DEVICE void scale_read_named(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536 && SLOT >= 0 && SLOT < 3, "ds offset / slot");
  if constexpr (SLOT == 0) asm volatile("ds_read_b32 a216, %0 offset:%1" ::"v"(addr), "i"(OFF) : "a216");
  if constexpr (SLOT == 1) asm volatile("ds_read_b32 a217, %0 offset:%1" ::"v"(addr), "i"(OFF) : "a217");
  if constexpr (SLOT == 2) asm volatile("ds_read_b32 a218, %0 offset:%1" ::"v"(addr), "i"(OFF) : "a218");
}
template <int SLOT>
LX2V_DEVICE int scale_fetch() {
  int r;
  if constexpr (SLOT == 0) asm volatile("v_accvgpr_read_b32 %0, a216" : "=v"(r));
  if constexpr (SLOT == 1) asm volatile("v_accvgpr_read_b32 %0, a217" : "=v"(r));
  if constexpr (SLOT == 2) asm volatile("v_accvgpr_read_b32 %0, a218" : "=v"(r));
  return r;
}

template <int OFF>
LX2V_DEVICE int scale_read(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  int r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}

// two e4m3 values into the low / high half of a dword
LX2V_DEVICE uint32_t cvt_lo_op(float a, float b) {
  uint32_t r;
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
LX2V_DEVICE uint32_t cvt_hi_op(uint32_t lo, float a, float b) {
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(lo) : "v"(a), "v"(b));
  return lo;
}

// The softmax share of one MFMA gap, as two asm statements (the compiler separates single-instruction
// statements that depend on each other by wait states a lone wave pays four cycles for each):
// four exps of this gap's P dword ...
LX2V_DEVICE void exp4_op(float (&e)[4], float s0, float s1, float s2, float s3) {
  asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7"
               : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
               : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
}
// ... and the consumers of the previous gap's exps: lane row sum and the e4m3 dword (adds and
// conversions alternate; none reads a register written by the instruction before it except the sum).
LX2V_DEVICE uint32_t sum_cvt4_op(float& ls, const float (&e)[4]) {
  uint32_t pk;
  asm volatile("v_add_f32 %0, %0, %2\n\tv_cvt_pk_fp8_f32 %1, %2, %3\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\t"
               "v_cvt_pk_fp8_f32 %1, %4, %5 op_sel:[0,0,1]\n\tv_add_f32 %0, %0, %5"
               : "+v"(ls), "=&v"(pk)
               : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
  return pk;
}

// ---- LIN: P without an exp ----
// An e4m3 code read as an integer is 8 (exponent + 7) + mantissa: a piecewise linear image of
// 8 (log2 + 7), with the subnormal codes continuing the line.  With the scores delivered by the QK
// MFMAs as 8 (s - m) + 56 (the Q scale bytes raised by 3, the accumulator init -8 m + 56), P is one
// saturating float -> byte conversion per score: 2^x with the mantissa taken linearly, at most 8.6 %
// above 2^x and 4.3 % on average -- a common factor that the row sums, taken over the same codes,
// divide out; what is left is an error of +-0.35 code that depends on the fraction of x, beside the
// +-0.5 code of the rounding every e4m3 P carries.
LX2V_DEVICE uint32_t cvt4_u8_op(float s0, float s1, float s2, float s3) {
  uint32_t pk;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0\n\tv_cvt_pk_u8_f32 %0, %2, 1, %0\n\tv_cvt_pk_u8_f32 %0, %3, 2, %0\n\tv_cvt_pk_u8_f32 %0, %4, 3, %0"
               : "=&v"(pk)
               : "v"(s0), "v"(s1), "v"(s2), "v"(s3)
               : F8_RING_CLOBBERS);
  return pk;
}
// The gap forms: one or two P dwords and the or of the tile's codes taken over two dwords at a time, as ONE statement (between two
// statements of which the second reads what the first wrote the compiler puts a wait state of its own).
LX2V_DEVICE uint32_t cvt4_u8_or3_op(uint32_t& acc, uint32_t other, float s0, float s1, float s2, float s3) {
  uint32_t pk;
  asm volatile("v_cvt_pk_u8_f32 %0, %3, 0, 0\n\tv_cvt_pk_u8_f32 %0, %4, 1, %0\n\tv_cvt_pk_u8_f32 %0, %5, 2, %0\n\tv_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
               "v_or3_b32 %1, %1, %0, %2"
               : "=&v"(pk), "+v"(acc)
               : "v"(other), "v"(s0), "v"(s1), "v"(s2), "v"(s3)
               : F8_RING_CLOBBERS);
  return pk;
}
LX2V_DEVICE void cvt8_u8_or_op(uint32_t& pa, uint32_t& pb, uint32_t& acc, float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
  asm volatile("v_cvt_pk_u8_f32 %0, %3, 0, 0\n\tv_cvt_pk_u8_f32 %1, %7, 0, 0\n\tv_cvt_pk_u8_f32 %0, %4, 1, %0\n\tv_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
               "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\tv_cvt_pk_u8_f32 %1, %9, 2, %1\n\tv_cvt_pk_u8_f32 %0, %6, 3, %0\n\tv_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
               "v_or3_b32 %2, %2, %0, %1"
               : "=&v"(pa), "=&v"(pb), "+v"(acc)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
               : F8_RING_CLOBBERS);
}
// Row sums on the MFMA pipe: lsum (accumulator file) += SEL x P^T, SEL a constant 16 x 128 operand
// whose rows 0-7 are ones on k-blocks 0 and 2 and rows 8-15 ones on k-blocks 1 and 3.  P is laid out
// for the 32x32x64 B operand (lane = row mod 32, k-block = lane / 32), which this shape reads as
// column lane & 15, k-block lane / 16: lanes 0-31 end up with the sum of row lane & 15, lanes 32-63
// with that of row 16 + (lane & 15), in all four registers.
LX2V_DEVICE void mfma_rowsum(f32x4& lsum, const i32x8& sel, const i32x8& pf, int unit) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(lsum) : "a"(sel), "v"(pf), "v"(unit) : F8_RING_CLOBBERS);
}
constexpr float kLinOne = 56.0f;       // code of P = 1
constexpr uint32_t kLinMaxTileSumBits = 0x43800000u;  // 256.0f: a tile that adds less than this to a row's sum holds no P at or above 2^8 (e4m3 ends at 448)

template <bool SHORT_KV, bool LIN>
__global__ __launch_bounds__(F8_THREADS) void attn_fwd_fp8_w4_kernel(const Fp8AttnParams p, const RoundSync rs) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  round_enter(rs);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const uint32_t wid_u = __builtin_amdgcn_readfirstlane(wid);

  const int qb_wg = blockIdx.x % p.nqb;
  const int head = blockIdx.x / p.nqb;
  const int q0 = qb_wg * QROWS + wid * 64;  // block qb covers rows q0 + 32 qb .. + 31

  // ---- Q fragments (B operand of S^T), parked in accumulator registers ----
  i32x8 qf[2][2];  // [qb][ks]
  int qsc[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow_c = min(q0 + 32 * qb + r32, p.sq - 1);
    const uint8_t* qp = p.q8 + ((int64_t)qrow_c * p.heads + head) * HD + 16 * h;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(qp + 64 * ks), b = *reinterpret_cast<const u32x4*>(qp + 64 * ks + 32);
      qf[qb][ks] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
    const uint32_t sc = *reinterpret_cast<const uint32_t*>(p.qs + ((int64_t)qrow_c * p.heads + head) * 4);
    qsc[qb] = (int)(((sc >> (8 * h)) & 0xffu) | (((sc >> (8 * (2 + h))) & 0xffu) << 8));  // byte ks = scale of block 2 ks + h
    if constexpr (LIN) qsc[qb] += 0x0303;  // scores times 8 (the scale bytes are exponents; Q scales sit far below 252)
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+a"(qf[qb][ks]));

  // ---- staging by LDS-DMA.  Tile t of this head starts at byte t * TILE_BLOB of the buffer; a wave
  // instruction copies 16 bytes per active lane to LDS byte M0 + 16 lane.  Per tile part a wave moves
  // two full KiB of rows (KiB w and w + 4 of the eight) and its share of the scales (K: 8 lanes, V: 4).
  const char* kv_base = reinterpret_cast<const char*>(p.kv) + (int64_t)head * p.tiles * TILE_BLOB;
  u32x4 kv_rs;
  {
    const uint64_t a = reinterpret_cast<uint64_t>(kv_base);
    kv_rs[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    kv_rs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));  // stride 0: raw buffer
    kv_rs[2] = __builtin_amdgcn_readfirstlane((uint32_t)((int64_t)p.tiles * TILE_BLOB));  // loads at or past it return 0
    kv_rs[3] = 0x00020000u;
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds);
  const uint32_t m0_rows = lds_base + 1024 * wid_u, m0_ksc = lds_base + 128 * wid_u, m0_vsc = lds_base + 64 * wid_u;
  const uint32_t src_rows = 1024 * wid_u, src_ksc = K_SCALE + 128 * wid_u, src_vsc = V_SCALE + 64 * wid_u;
  int dma_voff = lane * 16;
  asm volatile("" : "+v"(dma_voff));
  auto dma_m0 = [&](uint32_t base, auto dstc) __attribute__((always_inline)) {
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(base), "i"(decltype(dstc)::value) : "memory", "scc");
  };
  auto dma_go = [&](uint32_t soff) __attribute__((always_inline)) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dma_voff), "s"(kv_rs), "s"(soff) : "memory");
  };
  // piece I of the six of (K tile tk -> K slot KSL, V tile tv -> V slot VSL)
  auto dma_piece_m0 = [&](auto ic, auto kslc, auto vslc) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value, KSL = decltype(kslc)::value, VSL = decltype(vslc)::value;
    if constexpr (I == 0) dma_m0(m0_rows, IntC<K_RING + KSL * K_PART>{});
    if constexpr (I == 1) dma_m0(m0_rows, IntC<K_RING + KSL * K_PART + 4096>{});
    if constexpr (I == 2) dma_m0(m0_ksc, IntC<K_RING + KSL * K_PART + K_SCALE>{});
    if constexpr (I == 3) dma_m0(m0_rows, IntC<V_RING + VSL * V_PART>{});
    if constexpr (I == 4) dma_m0(m0_rows, IntC<V_RING + VSL * V_PART + 4096>{});
    if constexpr (I == 5) dma_m0(m0_vsc, IntC<V_RING + VSL * V_PART + (V_SCALE - V_DATA)>{});
  };
  auto dma_piece_go = [&](auto ic, uint32_t k_soff, uint32_t v_soff) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I == 0) dma_go(k_soff + src_rows);
    if constexpr (I == 1) dma_go(k_soff + src_rows + 4096);
    if constexpr (I == 2 && !(F8W4_ABL & 64)) {
      if (lane < 8) dma_go(k_soff + src_ksc);
    }
    if constexpr (I == 3) dma_go(v_soff + V_DATA + src_rows);
    if constexpr (I == 4) dma_go(v_soff + V_DATA + src_rows + 4096);
    if constexpr (I == 5 && !(F8W4_ABL & 64)) {
      if (lane < 4) dma_go(v_soff + src_vsc);
    }
  };
  auto dma_piece = [&](auto ic, auto kslc, auto vslc, int tk, int tv) __attribute__((always_inline)) {
    dma_piece_m0(ic, kslc, vslc);
    asm volatile("s_nop 0");
    dma_piece_go(ic, (uint32_t)tk * TILE_BLOB, (uint32_t)tv * TILE_BLOB);
  };

  // ---- fragment read addresses (lane parts; slot, key block and channel block are immediates) ----
  // K (A operand of S^T): key = 32 kb + r32, chunks 4 ks + h and 4 ks + 2 + h (swizzled by (key >> 1) & 7)
  uint32_t k_rd[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      k_rd[ks][i] = lds_base + K_DATA + r32 * 128 + (((4 * ks + 2 * i + h) ^ ((r32 >> 1) & 7)) << 4);
      asm volatile("" : "+v"(k_rd[ks][i]));
    }
  uint32_t ks_rd = lds_base + K_SCALE + (r32 * 2 + h) * 4;
  // V^T (A operand of O^T): d = 32 db + r32, slots 32 h .. + 31 = chunks 2 h, 2 h + 1 (swizzled by (d >> 2) & 3).
  // The V ring's base is part of the lane address: ds offsets are 16-bit.
  uint32_t v_rd[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    v_rd[i] = lds_base + V_RING + r32 * 64 + (((2 * h + i) ^ ((r32 >> 2) & 3)) << 4);
    asm volatile("" : "+v"(v_rd[i]));
  }
  uint32_t vs_rd = lds_base + V_RING + (V_SCALE - V_DATA) + (h * 32 + r32) * 4;
  asm volatile("" : "+v"(ks_rd), "+v"(vs_rd));

  // Fragment F of a period whose tile is t (TM = t mod 6): even F = K(t+1) block kb = (F >> 1) & 1,
  // channel step ks = F >> 2; odd F = V(t-1) channel block db = F >> 1.  Ring slot F & 3.
  auto frag_read = [&](auto tmc, auto fc, auto halfc) __attribute__((always_inline)) {
    constexpr int TM = decltype(tmc)::value, F = decltype(fc)::value, HALF = decltype(halfc)::value;
    if constexpr (F8W4_ABL & 2) return;
    if constexpr ((F & 1) == 0) {
      constexpr int kb = (F >> 1) & 1, ks = F >> 2, slot = (TM + 1) % NSLOT;
      ring_read<(F & 3), HALF, K_RING + slot * K_PART + kb * 4096>(k_rd[ks][HALF]);
    } else {
      constexpr int db = F >> 1, slot = (TM + NSLOT - 1) % NSLOT;
      ring_read<(F & 3), HALF, slot * V_PART + db * 2048>(v_rd[HALF]);
    }
  };

  f32x16 o[2][4];   // [qb][db], accumulator file: o[qb][db][i] = O[q0 + 32 qb + r32][32 db + 8 (i >> 2) + 4 h + (i & 3)]
  f32x16 negm[2];   // -m of the lane's row of block qb in all 16 registers
  f32x16 s[2][2][2];  // [parity of the tile][kb][qb]: s[..][i] is key 32 kb + 8 (i >> 2) + 4 h + (i & 3) of the tile
  i32x8 pf[2][2];   // [parity][qb]: P of the tile, dword 4 kb + w = keys of s[kb][4 w .. 4 w + 3]
  float l_run[2] = {0.f, 0.f};
  int ksc[2] = {0, 0}, vsc = 0;
  int unit = UNIT_SCALE;
  asm volatile("" : "+v"(unit));
  // LIN: row sums in the accumulator file (see mfma_rowsum), their value before the tile in flight, the constant operand
  f32x4 lsum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float lprev[2] = {0.f, 0.f};
  i32x8 sel;
  if constexpr (LIN) {
    const int one4 = (((lane & 15) < 8) == (((lane >> 4) & 1) == 0)) ? 0x38383838 : 0;  // e4m3 1.0 in every byte, or nothing
#pragma unroll
    for (int i = 0; i < 8; ++i) sel[i] = one4;
    asm volatile("" : "+a"(sel), "+a"(lsum[0]), "+a"(lsum[1]));
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[qb][i] = LIN ? kLinOne : 0.f;
    asm volatile("" : "+v"(negm[qb]));
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qb][db][i] = 0.f;
      asm volatile("" : "+a"(o[qb][db]));
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
#pragma unroll
      for (int i = 0; i < 8; ++i) pf[par][qb][i] = 0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[par][kb][qb][i] = -1e30f;  // "tile -1": every key masked
    }
  }

  const int nt = (p.skv + KVB - 1) / KVB;

  auto mask_tile = [&](f32x16 (&sv)[2][2], int key0) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (key0 + 32 * kb + 8 * (i >> 2) + 4 * h + (i & 3) >= p.skv) {
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) sv[kb][qb][i] = -1e30f;
        }
  };

  // Rare path, entered before any P of tile t is used: move O, l, m, S(t) and the complete S(t+1) of
  // block qb to the row maximum of S(t) (never downwards, except for tile 0, whose maximum becomes m
  // whatever its sign) and redo P(t).  MFMAs on these registers may have been issued a few cycles
  // ago and others follow right after, hence the explicit wait states.
  auto fixup = [&](auto qbc, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2], i32x8& pc, bool first, float& ls) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+v"(sc[0][qb]), "+v"(sc[1][qb]), "+v"(sn[0][qb]), "+v"(sn[1][qb]));
    float mx = sc[0][qb][0];
#pragma unroll
    for (int i = 1; i < 32; ++i) mx = fmaxf(mx, sc[i >> 4][qb][i & 15]);
    mx = half_swap_max(mx);
    if constexpr (LIN) mx -= kLinOne;  // scores are 8 (s - m) + 56 here, delta is in the same eighths
    const float delta = first ? mx : fmaxf(mx, 0.f);
    const float alpha = __builtin_amdgcn_exp2f(LIN ? -0.125f * delta : -delta);
    l_run[qb] *= alpha;
    if constexpr (LIN) {  // the row sums go back to their value before this tile, scaled; the tile is added again below
      const int lsum_row = (threadIdx.x & 15) + ((threadIdx.x >> 1) & 16);  // the row whose sum this lane's lsum holds (a lane of the same h: it has that row's alpha)
      const float base = lprev[qb] * __shfl(alpha, lsum_row, 64);
      asm volatile("s_nop 7\n\ts_nop 7" : "+a"(lsum[qb]));
      lsum[qb] = f32x4{base, base, base, base};
      asm volatile("" : "+a"(lsum[qb]));
    }
    static_for<0, 4>([&](auto dbc) __attribute__((always_inline)) {  // one tuple at a time through the architectural file
      constexpr int db = decltype(dbc)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qb][db][i] *= alpha;
      asm volatile("" : "+a"(o[qb][db]));
    });
    const float nm = negm[qb][0] - delta;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[qb][i] = nm;
    ls = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sc[kb][qb][4 * w + j] -= delta;
          sn[kb][qb][4 * w + j] -= delta;
          if constexpr (!LIN) {
            e[j] = __builtin_amdgcn_exp2f(sc[kb][qb][4 * w + j]);
            ls += e[j];
          }
        }
        if constexpr (LIN) {
          pc[4 * kb + w] = (int)cvt4_u8_op(sc[kb][qb][4 * w], sc[kb][qb][4 * w + 1], sc[kb][qb][4 * w + 2], sc[kb][qb][4 * w + 3]);
        } else {
          int pk = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], 0, false);
          pc[4 * kb + w] = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], pk, true);
        }
      }
    asm volatile("s_nop 1" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]), "+v"(negm[qb]), "+v"(pc), "+v"(sn[0][qb]), "+v"(sn[1][qb]));
    if constexpr (LIN) {
      mfma_rowsum(lsum[qb], sel, pc, unit);
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(lsum[qb]));
      lprev[qb] = lsum[qb][0];
    }
  };

  // One period.  TM = t mod 6 fixes every buffer: S / P parity TM & 1, K slot (TM + 1) % 6, V slot
  // (TM + 5) % 6.  Periods outside the real tiles (t = -1 before the first, t = nt at the end) run
  // the same code with every key masked.
  auto period = [&](auto tmc, int t) __attribute__((always_inline)) {
    constexpr int TM = decltype(tmc)::value, PAR = TM & 1;
    f32x16 (&Sc)[2][2] = s[PAR];       // S(t): exponentiated in this period
    f32x16 (&Sn)[2][2] = s[PAR ^ 1];   // S(t+1): accumulated in this period
    i32x8 (&Pc)[2] = pf[PAR];          // P(t): produced
    i32x8 (&Pp)[2] = pf[PAR ^ 1];      // P(t-1): consumed
    float ls[2] = {0.f, 0.f};
    uint32_t orv = 0;  // LIN: the P dwords of the tile or-ed together
    asm volatile("" : "+v"(orv));
    float lnow[2] = {0.f, 0.f};  // LIN: the row sums with the tile added, and whether the tile broke e4m3's range (per lane)
    uint32_t broke = 0;
    float e[2][4];  // exps of the dword of this gap / of the previous one
    if (__builtin_expect((t + 1) * KVB > p.skv, 0)) mask_tile(Sc, t * KVB);  // wave-uniform: the last real tile when ragged, and the tile past the end (kept out of the fall-through path)
    int ksc_next[2] = {0, 0}, vsc_next = 0;
    // tiles staged in this period (past the last tile of the image: the last one again, never used)
    const uint32_t k_soff = (uint32_t)min(t + 7, p.tiles - 1) * TILE_BLOB, v_soff = (uint32_t)min(t + 5, p.tiles - 1) * TILE_BLOB;
    static_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
      constexpr int F = decltype(fc)::value;
      // Tile barrier, in front of step 5: K(t+2) and V(t) (staged five periods ago) have landed; after
      // the barrier that holds for every wave, and the first reads of them follow in this step.
      // Every read of K(t+1) and V(t-1) -- the slots refilled right behind this barrier -- precedes
      // it in each wave's instruction stream (the last ones are the fragment requests of steps 3
      // and 4): the LDS queue serves them before the staged data, which arrives a memory round
      // trip later.
      if constexpr (F == 5 && !(F8W4_ABL & 8)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((4 * DMA_PER_PERIOD) & 15) | (((4 * DMA_PER_PERIOD) >> 4) << 14));  // vmcnt(24)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // fragment F has landed: reads issued behind it are those of fragments F + 1 and F + 2 (two
      // each) and the scale reads that precede K00 (two) and V0 (one) of the next period
      if constexpr (!(F8W4_ABL & (2 | 32))) wait_lds<4 + (F == 0 ? 1 : F == 6 ? 2 : F == 7 ? 3 : 0)>();
      if constexpr (LIN) {  // the wait of step 0 covers the K scale dwords of this period, that of step 1 the V scales
        if constexpr (F == 0) {
          ksc[0] = scale_fetch<0>();
          ksc[1] = scale_fetch<1>();
          asm volatile("s_nop 1" : "+v"(ksc[0]), "+v"(ksc[1]));
        }
        if constexpr (F == 1) {
          vsc = scale_fetch<2>();
          asm volatile("s_nop 1" : "+v"(vsc));
        }
      }
      static_for<0, 2>([&](auto qbc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qbc)::value, G = 2 * F + qb;
        if constexpr ((F & 1) == 0) {
          constexpr int kb = (F >> 1) & 1, ks = F >> 2;
          mfma_qk<(F & 3), ks, LIN>(Sn[kb][qb], qf[qb][ks], negm[qb], ksc[kb], qsc[qb]);
        } else {
          mfma_pv<(F & 3), (F >> 1)>(o[qb][F >> 1], Pp[qb], vsc, unit);
        }
        // staging: the six pieces of K(t+7) and V(t+5) in the gaps of steps 5, 6, 7
        if constexpr (F >= 5 && !(F8W4_ABL & 4)) dma_piece_m0(IntC<2 * (F - 5) + qb>{}, IntC<(TM + 1) % NSLOT>{}, IntC<(TM + 5) % NSLOT>{});
        // the half of fragment F + 3 this gap requests (scale dwords of the next period's tiles first)
        if constexpr (F + 3 < 8) {
          frag_read(tmc, IntC<F + 3>{}, qbc);
        } else {
          if constexpr (!(F8W4_ABL & 2)) {
            if constexpr (LIN) {
              if constexpr (F == 5 && qb == 0) {
                scale_read_named<0, K_RING + ((TM + 2) % NSLOT) * K_PART>(ks_rd);
                scale_read_named<1, K_RING + ((TM + 2) % NSLOT) * K_PART + 256>(ks_rd);
              }
              if constexpr (F == 6 && qb == 0) scale_read_named<2, (TM % NSLOT) * V_PART>(vs_rd);
            } else {
              if constexpr (F == 5 && qb == 0) {
                ksc_next[0] = scale_read<K_RING + ((TM + 2) % NSLOT) * K_PART>(ks_rd);
                ksc_next[1] = scale_read<K_RING + ((TM + 2) % NSLOT) * K_PART + 256>(ks_rd);
              }
              if constexpr (F == 6 && qb == 0) vsc_next = scale_read<(TM % NSLOT) * V_PART>(vs_rd);
            }
          }
          frag_read(IntC<(TM + 1) % NSLOT>{}, IntC<F - 5>{}, qbc);
        }
        if constexpr (F >= 5 && !(F8W4_ABL & 4)) dma_piece_go(IntC<2 * (F - 5) + qb>{}, k_soff, v_soff);
        // LIN: the sixteen P dwords are conversions of the scores, two per gap in gaps 0-3 (block 0) and one
        // per gap in gaps 4-11 (block 1), which keeps them out of the gaps that also stage; each block's
        // row-sum MFMA follows a gap later, behind that gap's VALU work (right behind the gap's own MFMA
        // it would hold the wave until that one has left the pipe).
        if constexpr (LIN && !(F8W4_ABL & 1)) {
          if constexpr (G < 4) {  // dwords 2 G and 2 G + 1 of block 0: the same key block, adjacent score quads
            constexpr int dkb = G >> 1, dw = 2 * (G & 1);
            uint32_t pa, pb;
            cvt8_u8_or_op(pa, pb, orv, Sc[dkb][0][4 * dw], Sc[dkb][0][4 * dw + 1], Sc[dkb][0][4 * dw + 2], Sc[dkb][0][4 * dw + 3],
                          Sc[dkb][0][4 * dw + 4], Sc[dkb][0][4 * dw + 5], Sc[dkb][0][4 * dw + 6], Sc[dkb][0][4 * dw + 7]);
            Pc[0][2 * G] = (int)pa;
            Pc[0][2 * G + 1] = (int)pb;
          } else if constexpr (G < 12) {  // one dword of block 1 per gap; the or takes two dwords at a time, in every second gap
            constexpr int D = G - 4, dkb = D >> 2, dw = D & 3;
            if constexpr ((D & 1) == 0) Pc[1][D] = (int)cvt4_u8_op(Sc[dkb][1][4 * dw], Sc[dkb][1][4 * dw + 1], Sc[dkb][1][4 * dw + 2], Sc[dkb][1][4 * dw + 3]);
            else Pc[1][D] = (int)cvt4_u8_or3_op(orv, (uint32_t)Pc[1][D - 1], Sc[dkb][1][4 * dw], Sc[dkb][1][4 * dw + 1], Sc[dkb][1][4 * dw + 2], Sc[dkb][1][4 * dw + 3]);
          }
          if constexpr (G == 5) mfma_rowsum(lsum[0], sel, Pc[0], unit);   // (gaps 4 / 13 instead: the same time)
          if constexpr (G == 12) mfma_rowsum(lsum[1], sel, Pc[1], unit);
          // The overflow check of the tile, in a gap that has nothing else to do: what the tile added to each row's sum.
          // The later row-sum MFMA was issued two gaps ago and two 16-pass MFMAs have entered the pipe behind it, each of
          // which waits for its predecessor to leave it: no wait states needed.  A P at or above 2^8 -- e4m3 ends at 448,
          // the code after it is NaN, saturated conversions read as negative numbers -- shows as an increase of at
          // least 2^8, as a NaN, or as a code with the sign bit.  Only the vote and the branch are left for the period's end.
          if constexpr (G == 14) {
            asm volatile("" : "+a"(lsum[0]), "+a"(lsum[1]));
            lnow[0] = lsum[0][0];
            lnow[1] = lsum[1][0];
            // (on the bit patterns: this file is compiled with "no NaNs", under which !(x < c) may become c <= x)
            const uint32_t d0 = __float_as_uint(lnow[0] - lprev[0]) & 0x7fffffffu, d1 = __float_as_uint(lnow[1] - lprev[1]) & 0x7fffffffu;
            broke = (d0 >= kLinMaxTileSumBits || d1 >= kLinMaxTileSumBits || (orv & 0x80808080u) != 0) ? 1u : 0u;
            asm volatile("" : "+v"(broke), "+v"(lnow[0]), "+v"(lnow[1]));
          }
        }
        // softmax share of gap G: exps of dword G, consumers of dword G - 1
        if constexpr (!LIN && !(F8W4_ABL & 1)) {
          constexpr int dq = G >> 3, dkb = (G >> 2) & 1, dw = G & 3;
          if constexpr (F8W4_ABL & 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[G & 1][j] = Sc[dkb][dq][4 * dw + j];
          } else {
            exp4_op(e[G & 1], Sc[dkb][dq][4 * dw], Sc[dkb][dq][4 * dw + 1], Sc[dkb][dq][4 * dw + 2], Sc[dkb][dq][4 * dw + 3]);
          }
          if constexpr (F8W4_ABL & 128) {
            asm volatile("" ::"v"(e[(G + 1) & 1][0]), "v"(e[(G + 1) & 1][1]), "v"(e[(G + 1) & 1][2]), "v"(e[(G + 1) & 1][3]));
          } else if constexpr (G > 0) {
            constexpr int pq = (G - 1) >> 3, pd = (G - 1) & 7;
            Pc[pq][pd] = (int)sum_cvt4_op(ls[pq], e[(G - 1) & 1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if constexpr (!LIN) {
      ksc[0] = ksc_next[0];
      ksc[1] = ksc_next[1];
      vsc = vsc_next;
    }
    // ---- consumers of the last dword, then the overflow check of tile t, before any P(t) is used ----
    if constexpr (!LIN && !(F8W4_ABL & 1)) {
      ls[1] += (e[1][0] + e[1][1]) + (e[1][2] + e[1][3]);
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(e[1][0], e[1][1], 0, false);
      Pc[1][7] = __builtin_amdgcn_cvt_pk_fp8_f32(e[1][2], e[1][3], pk, true);
    }
    asm volatile("" : "+v"(Pc[0]), "+v"(Pc[1]), "+v"(ls[0]), "+v"(ls[1]));
    const bool first = t == 0;  // tile 0: m becomes its true row maximum
    if constexpr (LIN) {
      const float l0 = lnow[0], l1 = lnow[1];
      const bool over = __any(broke != 0);
      if (!(F8W4_ABL & 16) && __builtin_expect(first || over, 0)) {
        static_for<0, 2>([&](auto qbc) __attribute__((always_inline)) { fixup(qbc, Sc, Sn, Pc[decltype(qbc)::value], first, ls[decltype(qbc)::value]); });
      } else {
        lprev[0] = l0;
        lprev[1] = l1;
      }
    } else {
      const int over = __any(!(fmaxf(ls[0], ls[1]) <= kMaxLaneSum));  // (one branch for "first or over")
      if (!(F8W4_ABL & 16) && __builtin_expect((int)first | over, 0)) {
        static_for<0, 2>([&](auto qbc) __attribute__((always_inline)) { fixup(qbc, Sc, Sn, Pc[decltype(qbc)::value], first, ls[decltype(qbc)::value]); });
      }
      add_op(l_run[0], ls[0]);   // (single adds: the compiler's v_pk_add_f32 costs a lone wave more than two of them)
      add_op(l_run[1], ls[1]);
    }
  };

  // ---- prologue: K(0..5) and V(0..3) staged; the V slots of the tiles "-2" and "-1" (4 and 5) zeroed
  // by hand (their P is 0, and 0 x whatever the slot held before must not be NaN) ----
  for (int i = tid; i < 2 * V_PART / 16; i += F8_THREADS) *reinterpret_cast<u32x4*>(lds + V_RING + 4 * V_PART + i * 16) = u32x4{0, 0, 0, 0};
  if constexpr (!(F8W4_ABL & 4)) {
    static_for<0, 6>([&](auto tc) __attribute__((always_inline)) {
      constexpr int T = decltype(tc)::value;
      static_for<0, 3>([&](auto ic) __attribute__((always_inline)) { dma_piece(ic, IntC<T>{}, IntC<0>{}, min(T, p.tiles - 1), 0); });
    });
    static_for<0, 4>([&](auto tc) __attribute__((always_inline)) {
      constexpr int T = decltype(tc)::value;
      static_for<3, 6>([&](auto ic) __attribute__((always_inline)) { dma_piece(ic, IntC<0>{}, IntC<T>{}, 0, min(T, p.tiles - 1)); });
    });
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): everything staged and zeroed by this wave is in LDS
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // fragments 0..2 and the scales of the first period (t = -1: K(0) in slot 0, "V(-2)" in slot 4)
  if constexpr (LIN) {
    scale_read_named<0, K_RING>(ks_rd);
    scale_read_named<1, K_RING + 256>(ks_rd);
  } else if constexpr (!(F8W4_ABL & 2)) {
    ksc[0] = scale_read<K_RING>(ks_rd);
    ksc[1] = scale_read<K_RING + 256>(ks_rd);
  }
  frag_read(IntC<5>{}, IntC<0>{}, IntC<0>{});
  frag_read(IntC<5>{}, IntC<0>{}, IntC<1>{});
  if constexpr (LIN) scale_read_named<2, 4 * V_PART>(vs_rd);
  else if constexpr (!(F8W4_ABL & 2)) vsc = scale_read<4 * V_PART>(vs_rd);
  frag_read(IntC<5>{}, IntC<1>{}, IntC<0>{});
  frag_read(IntC<5>{}, IntC<1>{}, IntC<1>{});
  frag_read(IntC<5>{}, IntC<2>{}, IntC<0>{});
  frag_read(IntC<5>{}, IntC<2>{}, IntC<1>{});

  // Periods t = -1 .. nt in a fixed rotation of six, so that every LDS offset and register is static;
  // the last one holds PV of the last tile.
  for (int t = -1;; t += 6) {
    period(IntC<5>{}, t);
    if (t == nt) break;
    period(IntC<0>{}, t + 1);
    if (t + 1 == nt) break;
    period(IntC<1>{}, t + 2);
    if (t + 2 == nt) break;
    period(IntC<2>{}, t + 3);
    if (t + 3 == nt) break;
    period(IntC<3>{}, t + 4);
    if (t + 4 == nt) break;
    period(IntC<4>{}, t + 5);
    if (t + 5 == nt) break;
  }
  // nothing may still be writing this workgroup's LDS or the fragment ring when it ends
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)

  round_leave(rs);

  // ---- normalise and store ----
  // (lane and row numbers once more from a register the loop keeps anyway: kept alive from the
  // prologue they cost the loop registers it does not have)
  const int lane_e = dma_voff >> 4, r32_e = lane_e & 31, h_e = lane_e >> 5;
  const int q0_e = qb_wg * QROWS + (int)wid_u * 64;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(o[qb][0]), "+a"(o[qb][1]), "+a"(o[qb][2]), "+a"(o[qb][3]));
    float lrow;
    if constexpr (LIN) {
      asm volatile("s_nop 7\n\ts_nop 7" : "+a"(lsum[qb]));
      lrow = __shfl(lsum[qb][0], r32_e < 16 ? r32_e : r32_e + 16, 64);
    } else {
      lrow = half_swap_sum(l_run[qb]);
    }
    const float inv = 1.0f / lrow;
    const int qrow = q0_e + 32 * qb + r32_e;
    if (qrow < p.sq) {
      uint16_t* op = p.out + (int64_t)qrow * p.ldo + head * HD + 4 * h_e;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk = {pack2(o[qb][db][4 * g] * inv, o[qb][db][4 * g + 1] * inv), pack2(o[qb][db][4 * g + 2] * inv, o[qb][db][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g) = pk;
        }
    }
  }
}

}  // namespace

int lx2v_attn_fp8_launch_w4(const Fp8AttnParams& p, hipStream_t st) {
  // workgroups of one round sweep a head's image together (see attn_w4_common.h)
  if (lx2v_options().attn_fp8_p == 1)
    return launch_with_rounds(attn_fwd_fp8_w4_kernel<true, true>, attn_fwd_fp8_w4_kernel<false, true>, F8_THREADS, p, (int64_t)p.tiles * TILE_BLOB, (int64_t)p.nqb * p.heads, st,
                              "attn_fwd_fp8(w4, linear P)");
  return launch_with_rounds(attn_fwd_fp8_w4_kernel<true, false>, attn_fwd_fp8_w4_kernel<false, false>, F8_THREADS, p, (int64_t)p.tiles * TILE_BLOB, (int64_t)p.nqb * p.heads, st,
                            "attn_fwd_fp8(w4)");
}
