// The chain pipeline for automata that fit the LDS whole (compact_tables.h): the per-byte DFA walk
// of ahomatch.cl:50-77 with every lookup served by the CU's own LDS, and the ordered compaction
// behind it (databuf.c:648-651, compactarray.cl:49-55 layout).  gfx950 only.
//
//   k_lds_walk     persistent workgroups of 16 waves, one per CU, the automaton image (<= 160 KiB)
//                  copied to LDS once per launch.  A wave takes tiles of C * 64 chains of S = 64 bytes;
//                  every lane walks its C chains, each from hb = L - 1 bytes (rounded up to 16) in
//                  front of it: the state of an Aho-Corasick automaton only remembers the last L - 1
//                  bytes, so from the chain's own first byte on the walk is the serial one (halo
//                  mode, scan.hip).  Per byte: the class of the byte, the state's 8-byte record, the
//                  row cell the record points at -- three LDS reads, no global memory, no branch, no
//                  loop: a lane whose record defers to its fail state's (1 % of the steps on the
//                  sentiment corpus) goes on in that state and takes the same byte again with the next
//                  step (step2_asm).  A transition into a final state stores {state code, step} in the
//                  lane's own list of the tile's staging area, lists of the 64 lanes interleaved so
//                  that the k-th records lie side by side.
//   k_lds_scatter_wide  per 2048 chains (512 threads x 4): scan of the counts on top of the totals of the
//                  tiles in front, records copied to their final, position-ordered cells with the
//                  pattern looked up, header and trailer cells.  Sized to run beside the NEXT stream's
//                  walk kernel (64 registers, 192 B of LDS).
// Both kernels take a GROUP of up to 16 batches of one size (acm_scan_batches_async): the image is
// copied once, the launch boundaries are paid once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "acm_internal.h"
#include "compact_tables.h"
#include "device_dfa.h"
#include "lds_walk.h"

namespace {

constexpr int kWalkBlock = 1024;               // 16 waves: the whole CU (the image takes all of its LDS)
constexpr int kWalkWaves = kWalkBlock / 64;
constexpr uint32_t kChainBytes = 64, kLogChain = 6;
constexpr uint32_t kMaxHaloGroups = 2;         // hb <= 32: patterns of up to 33 bytes
constexpr uint32_t kGroups = kChainBytes / 16 + kMaxHaloGroups;   // 16-byte groups a lane holds per chain
constexpr uint32_t kMaxGroup = 16;
constexpr int kChains = 2;                     // chains per lane: 93 VGPRs, 16 waves per CU (four would spill at the 128 a 1024-thread workgroup gets)

struct LdsBatch {
	const uint8_t *text;
	uint32_t *stage;          // [tiles][C][S][64] {code | step << 16}
	uint8_t *cnt;             // [chains] records of each chain
	uint32_t *tile_total;     // [tiles]
	uint32_t *misc;           // [0] code of the state after the last byte, [2] path marker
	int32_t *pat_plane, *off_plane;
	uint32_t init_code, drop_before;
	int32_t off_shift;
	uint32_t plane_capacity, report_state;
	const uint32_t *init_ptr;   // not null: the code of the state to start in is there (handed over on the device)
};

struct LdsGroup {
	const uint4 *image;       // the LDS image, image16 uint4s
	uint32_t image16;
	uint32_t off_rec;         // byte offset of the records in the image
	uint32_t root_code, final_code;   // the root's code (= off_rec / 8); codes from final_code on are final states
	const int32_t *out;       // [cid] head pattern of a final state
	const uint32_t *cid2ref;  // [cid] reference id
	uint32_t n, n_chains, n_tiles, hb;
	uint32_t count;
	LdsBatch b[kMaxGroup];
};

// what the walk needs to know of the code space (compact_tables.h)
struct Codes {
	uint32_t off_rec;   // byte address of the records
	uint32_t root;      // code of the root = off_rec / 8
	uint32_t fin;       // codes from here on are final states
};

// One transition by the book (the whole deferral chain followed on the spot): the edge tiles, the drains below,
// the lanes that cannot fall further behind.
__device__ __forceinline__ uint32_t next_code(const uint8_t *lds, uint32_t off_rec, uint32_t e, uint32_t byte)
{
	const uint32_t cls2 = lds[byte];   // 2 * class
	uint32_t at = (e & 0xFFFFu) << 3, out = 0;   // the code is the record's address / 8 (codes in flight carry other fields above it)
	for (bool more = true; more;) {
		const uint2 q = *(const uint2 *)(lds + at);
		more = false;
		if (((q.x >> 16) & 0xFFu) == cls2) {
			out = q.x & 0xFFFFu;
		} else if ((q.x >> 24) == cls2) {
			out = q.y & 0xFFFFu;
		} else if (q.y < (acm::kCompactSideBase << 16)) {
			out = *(const uint16_t *)(lds + (q.y >> 16) + cls2);
		} else {
			more = true;
			at = off_rec + (((q.y >> 16) - acm::kCompactSideBase) << 3);   // the fail state's record
		}
	}
	return out;
}

// One byte for C chains.  e[] are state codes (the record's byte address / 8; compact_tables.h).
//   LIVE   the step may lie outside the chain's text (tile at either end of the text): such steps
//          leave the state alone;  EMIT  the step belongs to the chain itself, not to its halo.
template <int C, int K, bool LIVE, bool EMIT>
__device__ __forceinline__ void step_all(const uint8_t *lds, uint32_t off_rec, uint32_t final_code, const uint4 (&w)[C], uint32_t (&e)[C],
    uint32_t (&so)[C], uint32_t *stage, uint32_t stepbits, const int32_t (&lo)[C], const int32_t (&hi)[C],
    const int32_t (&keep)[C], int32_t j)
{
	uint32_t nxt[C];
#pragma unroll
	for (int c = 0; c < C; c++) {
		const uint32_t d = (K < 4) ? w[c].x : (K < 8) ? w[c].y : (K < 12) ? w[c].z : w[c].w;
		nxt[c] = next_code(lds, off_rec, e[c], (d >> (8 * (K & 3))) & 0xFFu);
	}
#pragma unroll
	for (int c = 0; c < C; c++) {
		if (LIVE && (j < lo[c] || j >= hi[c]))
			nxt[c] = e[c];   // in front of where this chain's walk starts, or past the end of the text: freeze
		e[c] = nxt[c];
		if (EMIT) {
			const bool hit = e[c] >= final_code && (!LIVE || (j >= keep[c] && j < hi[c]));
			if (hit) {
				stage[so[c]] = e[c] | stepbits;
				so[c] += 64;
			}
		}
	}
}

// The step for two chains, written out.  (The compiler's version of step_all spends more on moving
// exec masks around its deferral loop and hit stores than on the walk: 27 vector + 24 scalar
// instructions per byte and chain on the sentiment workload, and the loop -- entered by 59 % of the
// steps because SOME lane of 128 defers -- cost 18 of the kernel's 57 us.)  Here a lane whose record
// defers to its fail state's does not hold the wave up: it goes on IN the fail state and takes the same
// byte again with the next step -- it falls a byte behind.  nd = 4 - (bytes behind): the byte a lane
// looks at is picked by v_perm from the dword of the step and the one in front of it, selector nd + r.
// A lane can be four bytes behind; one that would fall further is left to the caller (slowA / slowB:
// once in a few thousand chains).  The drains at the end of the halo and of the chain bring everybody
// level again.  v84..v87 hold the two records (fixed registers: the halves of a 64-bit asm operand cannot be named; low ones, so that they do not push the kernel's register count up); the two chains' LDS reads are interleaved and waited for
// by count.  EMIT: a final state entered (by a lane that did not defer) is stored as
// {code, step - bytes behind} in the lane's list, 256 bytes further for every record.
template <int R, bool EMIT>
__device__ __forceinline__ void step2_asm(uint32_t hiA, uint32_t loA, uint32_t hiB, uint32_t loB, uint32_t inA, uint32_t inB,
    uint32_t &eA, uint32_t &eB, uint32_t &ndA, uint32_t &ndB, uint32_t &soA, uint32_t &soB,
    uint32_t root_code, uint32_t final_code, uint32_t stepconst, uint32_t *stage, uint64_t &slowA, uint64_t &slowB)
{
	uint64_t m1A, m2A, m1B, m2B, moreA, moreB, sv;
	uint32_t t0, t1, t2, t3, clsA, clsB;
	const uint32_t c0 = acm::kCompactSideBase << 16, kfix = root_code - acm::kCompactSideBase, ffff = 0xFFFFu;
	// Nine vector instructions per chain and byte (+ four where records are emitted): byte select (2), record
	// address = low half of the code * 8 (v_mad_u32_u16: the codes in flight carry other fields of the
	// records in their upper halves), cell address = next16 + 2*class in one add, three compares into scalar pairs,
	// two selects.  (Compares into VCC + SDWA selects would give clean codes and a one-instruction record,
	// v_add3 -- and make the walk a third slower: every VCC hand-over between vector and scalar unit stalls
	// the wave.)
#define ACM_WALK(sel)                                                                                              \
	"v_add_u32 %[t0], " sel ", %[ndA]\n\t"                                                                      \
	"v_perm_b32 %[t0], %[hiA], %[loA], %[t0]\n\t"                                                               \
	"v_mad_u32_u16 %[t1], %[inA], 8, 0\n\t"                                                                     \
	"ds_read_u8 %[clsA], %[t0]\n\t"                                                                             \
	"ds_read_b64 v[84:85], %[t1]\n\t"                                                                           \
	"v_add_u32 %[t2], " sel ", %[ndB]\n\t"                                                                      \
	"v_perm_b32 %[t2], %[hiB], %[loB], %[t2]\n\t"                                                               \
	"v_mad_u32_u16 %[t3], %[inB], 8, 0\n\t"                                                                     \
	"ds_read_u8 %[clsB], %[t2]\n\t"                                                                             \
	"ds_read_b64 v[86:87], %[t3]\n\t"                                                                           \
	"s_waitcnt lgkmcnt(2)\n\t"                                                                                  \
	"v_add_u32_sdwa %[t0], %[clsA], v85 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
	"ds_read_u16 %[t0], %[t0]\n\t"                                                                              \
	"v_cmp_eq_u32_sdwa %[m1A], v84, %[clsA] src0_sel:BYTE_2 src1_sel:DWORD\n\t"                                 \
	"v_cmp_eq_u32_sdwa %[m2A], v84, %[clsA] src0_sel:BYTE_3 src1_sel:DWORD\n\t"                                 \
	"v_cmp_le_u32_e64 %[moreA], %[c0], v85\n\t"                                                                 \
	"s_waitcnt lgkmcnt(1)\n\t"                                                                                  \
	"v_add_u32_sdwa %[t2], %[clsB], v87 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
	"ds_read_u16 %[t2], %[t2]\n\t"                                                                              \
	"v_cmp_eq_u32_sdwa %[m1B], v86, %[clsB] src0_sel:BYTE_2 src1_sel:DWORD\n\t"                                 \
	"v_cmp_eq_u32_sdwa %[m2B], v86, %[clsB] src0_sel:BYTE_3 src1_sel:DWORD\n\t"                                 \
	"v_cmp_le_u32_e64 %[moreB], %[c0], v87\n\t"                                                                 \
	"s_mov_b64 %[slowA], 0\n\t"                                                                                 \
	"s_mov_b64 %[slowB], 0\n\t"                                                                                 \
	"s_waitcnt lgkmcnt(1)\n\t"                                                                                  \
	"v_cndmask_b32_e64 %[eA], %[t0], v85, %[m2A]\n\t"                                                           \
	"v_cndmask_b32_e64 %[eA], %[eA], v84, %[m1A]\n\t"                                                           \
	"s_or_b64 %[m1A], %[m1A], %[m2A]\n\t"                                                                       \
	"s_andn2_b64 %[moreA], %[moreA], %[m1A]\n\t"                                                                \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                  \
	"v_cndmask_b32_e64 %[eB], %[t2], v87, %[m2B]\n\t"                                                           \
	"v_cndmask_b32_e64 %[eB], %[eB], v86, %[m1B]\n\t"                                                           \
	"s_or_b64 %[m1B], %[m1B], %[m2B]\n\t"                                                                       \
	"s_andn2_b64 %[moreB], %[moreB], %[m1B]\n\t"                                                                \
	"s_or_b64 %[sv], %[moreA], %[moreB]\n\t"                                                                    \
	"s_cbranch_scc0 .Lnodefer%=\n\t"                                                                            \
	"v_cmp_eq_u32_e64 %[slowA], 0, %[ndA]\n\t"                                                                  \
	"v_cmp_eq_u32_e64 %[slowB], 0, %[ndB]\n\t"                                                                  \
	"s_and_b64 %[slowA], %[slowA], %[moreA]\n\t"                                                                \
	"s_and_b64 %[slowB], %[slowB], %[moreB]\n\t"                                                                \
	"s_andn2_b64 %[m1A], %[moreA], %[slowA]\n\t"                                                                \
	"s_andn2_b64 %[m1B], %[moreB], %[slowB]\n\t"                                                                \
	"v_lshrrev_b32 %[t1], 16, v85\n\t"                                                                          \
	"v_lshrrev_b32 %[t3], 16, v87\n\t"                                                                          \
	"v_add_u32 %[t1], %[kfix], %[t1]\n\t"                                                                       \
	"v_add_u32 %[t3], %[kfix], %[t3]\n\t"                                                                       \
	"v_cndmask_b32_e64 %[eA], %[eA], %[t1], %[m1A]\n\t"                                                         \
	"v_cndmask_b32_e64 %[eB], %[eB], %[t3], %[m1B]\n\t"                                                         \
	"v_subb_co_u32_e64 %[ndA], %[m2A], %[ndA], 0, %[m1A]\n\t"                                                   \
	"v_subb_co_u32_e64 %[ndB], %[m2B], %[ndB], 0, %[m1B]\n\t"                                                   \
	".Lnodefer%=:\n\t"
#define ACM_EMIT                                                                                                   \
	"v_cmp_le_u16_e64 %[m2A], %[fc], %[eA]\n\t"                                                                 \
	"v_cmp_le_u16_e64 %[m2B], %[fc], %[eB]\n\t"                                                                 \
	"s_andn2_b64 %[m2A], %[m2A], %[moreA]\n\t"                                                                  \
	"s_andn2_b64 %[m2B], %[m2B], %[moreB]\n\t"                                                                  \
	"s_mov_b64 %[sv], exec\n\t"                                                                                 \
	"s_mov_b64 exec, %[m2A]\n\t"                                                                                \
	"v_lshl_add_u32 %[t0], %[ndA], 16, %[sc]\n\t"                                                               \
	"v_and_or_b32 %[t0], %[eA], %[ffff], %[t0]\n\t"                                                             \
	"global_store_dword %[soA], %[t0], %[base]\n\t"                                                             \
	"v_add_u32 %[soA], 0x100, %[soA]\n\t"                                                                       \
	"s_mov_b64 exec, %[m2B]\n\t"                                                                                \
	"v_lshl_add_u32 %[t2], %[ndB], 16, %[sc]\n\t"                                                               \
	"v_and_or_b32 %[t2], %[eB], %[ffff], %[t2]\n\t"                                                             \
	"global_store_dword %[soB], %[t2], %[base]\n\t"                                                             \
	"v_add_u32 %[soB], 0x100, %[soB]\n\t"                                                                       \
	"s_mov_b64 exec, %[sv]\n\t"
#define ACM_OPS                                                                                                    \
	: [eA] "=&v"(eA), [eB] "=&v"(eB), [ndA] "+v"(ndA), [ndB] "+v"(ndB), [soA] "+v"(soA), [soB] "+v"(soB),        \
	  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [clsA] "=&v"(clsA), [clsB] "=&v"(clsB),    \
	  [m1A] "=&s"(m1A), [m2A] "=&s"(m2A), [m1B] "=&s"(m1B), [m2B] "=&s"(m2B), [moreA] "=&s"(moreA),             \
	  [moreB] "=&s"(moreB), [slowA] "=&s"(slowA), [slowB] "=&s"(slowB), [sv] "=&s"(sv)                          \
	: [hiA] "v"(hiA), [loA] "v"(loA), [hiB] "v"(hiB), [loB] "v"(loB), [inA] "v"(inA), [inB] "v"(inB),            \
	  [kfix] "s"(kfix), [c0] "s"(c0), [fc] "s"(final_code), [ffff] "s"(ffff), [sc] "s"(stepconst), \
	  [base] "s"(stage)                                                                                            \
	: "v84", "v85", "v86", "v87", "memory", "scc"
	// selector of v_perm: byte nd + r of {hi, lo} (lo = bytes 0..3), the other three bytes of the result zero
	if (EMIT) {
		if (R == 0) asm volatile(ACM_WALK("0x0c0c0c00") ACM_EMIT ACM_OPS);
		else if (R == 1) asm volatile(ACM_WALK("0x0c0c0c01") ACM_EMIT ACM_OPS);
		else if (R == 2) asm volatile(ACM_WALK("0x0c0c0c02") ACM_EMIT ACM_OPS);
		else asm volatile(ACM_WALK("0x0c0c0c03") ACM_EMIT ACM_OPS);
	} else {
		if (R == 0) asm volatile(ACM_WALK("0x0c0c0c00") ACM_OPS);
		else if (R == 1) asm volatile(ACM_WALK("0x0c0c0c01") ACM_OPS);
		else if (R == 2) asm volatile(ACM_WALK("0x0c0c0c02") ACM_OPS);
		else asm volatile(ACM_WALK("0x0c0c0c03") ACM_OPS);
	}
#undef ACM_OPS
#undef ACM_EMIT
#undef ACM_WALK
}

// byte K of the 16 a lane holds of each of its two chains (prev: the last dword of the group in front)
template <int K, bool EMIT>
__device__ __forceinline__ void step_pair(const uint8_t *lds, const Codes &k, const uint4 (&w)[2], const uint32_t (&prev)[2],
    uint32_t (&e)[2], uint32_t (&nd)[2], uint32_t (&so)[2], uint32_t *stage, uint32_t step, uint32_t lane)
{
	constexpr int Q = K >> 2, R = K & 3;
	const uint32_t hiA = Q == 0 ? w[0].x : Q == 1 ? w[0].y : Q == 2 ? w[0].z : w[0].w;
	const uint32_t loA = Q == 0 ? prev[0] : Q == 1 ? w[0].x : Q == 2 ? w[0].y : w[0].z;
	const uint32_t hiB = Q == 0 ? w[1].x : Q == 1 ? w[1].y : Q == 2 ? w[1].z : w[1].w;
	const uint32_t loB = Q == 0 ? prev[1] : Q == 1 ? w[1].x : Q == 2 ? w[1].y : w[1].z;
	const uint32_t inA = e[0], inB = e[1];   // (the step writes new registers: no copies)
	uint64_t slowA, slowB;
	step2_asm<R, EMIT>(hiA, loA, hiB, loB, inA, inB, e[0], e[1], nd[0], nd[1], so[0], so[1], k.root, k.fin, (step - 4u) << 16, stage,
	    slowA, slowB);
	if (slowA | slowB) {   // a lane four bytes behind deferred again: its transition by the book, no falling further behind
		const uint32_t in[2] = { inA, inB }, lw[2] = { loA, loB };
		const uint64_t slow[2] = { slowA, slowB };
#pragma unroll
		for (int c = 0; c < 2; c++)
			if ((slow[c] >> lane) & 1ull) {
				e[c] = next_code(lds, k.off_rec, in[c], (lw[c] >> (8 * R)) & 0xFFu);   // four behind: byte r of the dword in front
				if (EMIT && e[c] >= k.fin) {
					*(uint32_t *)((char *)stage + so[c]) = e[c] | ((step - 4u) << 16);
					so[c] += 256;
				}
			}
	}
}

// The lanes that are behind take the bytes they have not looked at yet -- the last 4 - nd of the group
// just walked (lastw: its last dword) -- by the book; afterwards every lane is level (nd = 4).
template <bool EMIT>
__device__ __forceinline__ void drain(const uint8_t *lds, const Codes &k, uint32_t lastw, uint32_t &e, uint32_t &nd, uint32_t &so,
    uint32_t *stage, uint32_t end_step)
{
	uint32_t d = 4u - nd;
	while (__builtin_amdgcn_ballot_w64(d != 0)) {
		if (d != 0) {
			e = next_code(lds, k.off_rec, e, (lastw >> (8 * (4u - d))) & 0xFFu);
			if (EMIT && e >= k.fin) {
				*(uint32_t *)((char *)stage + so) = e | ((end_step - d) << 16);
				so += 256;
			}
			d--;
		}
	}
	nd = 4;
}

// A whole tile (no end of the text, no shard halo inside it) of two chains per lane, the written-out step.
// NG: 16-byte groups a lane holds per chain (5: a halo of one group, 6: of two).
template <int NG>
__device__ __forceinline__ void walk_tile_fast(const LdsGroup &g, const LdsBatch &b, const uint8_t *lds, uint32_t wt, uint32_t lane)
{
	constexpr int C = 2;
	const uint32_t hb = g.hb;
	const Codes k = { g.off_rec, g.root_code, g.final_code };
	uint32_t e[C], so[C], so0[C], nd[C], chain[C], prev[C];
	// (wave-uniform, but derived from the thread index: pinned to scalar registers for the stores' base operand)
	const uint64_t sp = (uint64_t)(uintptr_t)(b.stage + (size_t)wt * (C * kChainBytes * 64));
	// (the builtin returns an int: without the casts a low half with its top bit set is sign-extended over the high one)
	const uint32_t sp_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sp >> 32));
	const uint32_t sp_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sp);
	uint32_t *stage = (uint32_t *)(uintptr_t)(((uint64_t)sp_hi << 32) | (uint64_t)sp_lo);
	uint4 p0[C], p1[C], p2[C], p3[C], p4[C], p5[C];
	const uint4 *text16 = (const uint4 *)b.text;
	const uint32_t groups = (kChainBytes + hb) >> 4, hg = hb >> 4;
#pragma unroll
	for (int c = 0; c < C; c++) {
		chain[c] = (wt * C + c) * 64 + lane;
		const uint32_t base = chain[c] << kLogChain;
		e[c] = k.root;
		nd[c] = 4;
		prev[c] = 0;
		so0[c] = so[c] = ((uint32_t)c * (kChainBytes * 64) + lane) * 4;   // byte offset into the tile's staging area
		// (the text's very first chain has no halo: its halo groups read the start of the text instead, and
		// what the walk makes of them is thrown away below)
		// (a chain behind the last of the text -- the text's last tile, when the chains are not a multiple of the
		// tile -- walks the text's last bytes instead; what it finds stays in the staging area, nobody counts it)
		const int32_t gl = (int32_t)((g.n + 15) >> 4) - 1;   // the last group of the text (whole or not: the buffer is padded to 16 bytes)
		const int32_t g0 = (int32_t)(base >> 4) - (int32_t)hg;
		p0[c] = text16[min(max(g0, 0), gl)];
		p1[c] = text16[min(max(g0 + 1, 0), gl)];
		p2[c] = text16[min(g0 + 2, gl)];
		p3[c] = text16[min(g0 + 3, gl)];
		p4[c] = groups > 4 ? text16[min(g0 + 4, gl)] : make_uint4(0, 0, 0, 0);
		if constexpr (NG > 5)
			p5[c] = groups > 5 ? text16[min(g0 + 5, gl)] : make_uint4(0, 0, 0, 0);
	}
	static_assert(kMaxHaloGroups == 2, "the first two groups may be halo");
#define ACM_PICK(gi)                                                                     \
	{                                                                                \
		const uint32_t gu = __builtin_amdgcn_readfirstlane(gi);                  \
		_Pragma("unroll") for (int c = 0; c < C; c++)                            \
		{                                                                        \
			if (gu == 0) w[c] = p0[c];                                       \
			else if (gu == 1) w[c] = p1[c];                                  \
			else if (gu == 2) w[c] = p2[c];                                  \
			else if (gu == 3) w[c] = p3[c];                                  \
			else if (NG <= 5 || gu == 4) w[c] = p4[c];                       \
			else w[c] = p5[c];                                               \
		}                                                                        \
	}
#define ACM_STEP(K, EM) step_pair<K, EM>(lds, k, w, prev, e, nd, so, stage, s0 + (K), lane)
#define ACM_STEPS(EM) ACM_STEP(0, EM); ACM_STEP(1, EM); ACM_STEP(2, EM); ACM_STEP(3, EM); ACM_STEP(4, EM); ACM_STEP(5, EM); \
	ACM_STEP(6, EM); ACM_STEP(7, EM); ACM_STEP(8, EM); ACM_STEP(9, EM); ACM_STEP(10, EM); ACM_STEP(11, EM); ACM_STEP(12, EM); \
	ACM_STEP(13, EM); ACM_STEP(14, EM); ACM_STEP(15, EM)
	uint4 w[C];
	w[0] = w[1] = make_uint4(0, 0, 0, 0);
#pragma nounroll
	for (uint32_t gi = 0; gi < hg; gi++) {   // the halo: only the state matters
		ACM_PICK(gi);
		const uint32_t s0 = 0;
		ACM_STEPS(false);
		prev[0] = w[0].w;
		prev[1] = w[1].w;
	}
	if (hg) {
		drain<false>(lds, k, w[0].w, e[0], nd[0], so[0], stage, 0);
		drain<false>(lds, k, w[1].w, e[1], nd[1], so[1], stage, 0);
	}
	if (chain[0] == 0)
		e[0] = b.init_ptr ? *b.init_ptr : b.init_code;   // the text's first chain starts at byte 0, in the carried-in state
#pragma nounroll
	for (uint32_t gi = hg; gi < groups; gi++) {
		ACM_PICK(gi);
		const uint32_t s0 = (gi - hg) * 16;   // step inside the chain of the group's first byte
		ACM_STEPS(true);
		prev[0] = w[0].w;
		prev[1] = w[1].w;
	}
	drain<true>(lds, k, w[0].w, e[0], nd[0], so[0], stage, kChainBytes);
	drain<true>(lds, k, w[1].w, e[1], nd[1], so[1], stage, kChainBytes);
#undef ACM_STEPS
#undef ACM_STEP
#undef ACM_PICK
	uint32_t total = 0;
#pragma unroll
	for (int c = 0; c < C; c++) {
		const uint32_t base = chain[c] << kLogChain;
		if (chain[c] < g.n_chains && base + kChainBytes > g.n) {
			// The chain the text ends inside (one lane of one wave per batch): the written-out walk took bytes
			// behind the end for text.  Once more by the book, byte by byte, over what there is.
			const uint32_t from = base > hb ? base - hb : 0u;
			uint32_t st = base > hb ? k.root : (b.init_ptr ? *b.init_ptr : b.init_code);
			so[c] = so0[c];
			for (uint32_t x = from; x < g.n; x++) {
				st = next_code(lds, k.off_rec, st, b.text[x]);
				if (x >= base && st >= k.fin) {
					*(uint32_t *)((char *)stage + so[c]) = st | ((x - base) << 16);
					so[c] += 256;
				}
			}
			e[c] = st;
		}
		const uint32_t kcount = (so[c] - so0[c]) >> 8;
		if (chain[c] < g.n_chains) {
			b.cnt[chain[c]] = (uint8_t)kcount;
			total += kcount;
			if (chain[c] == g.n_chains - 1)
				b.misc[0] = e[c] & 0xFFFFu;
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		total += __shfl_xor(total, o, 64);
	if (lane == 0)
		b.tile_total[wt] = total;
}

template <int C, bool LIVE, int NG = 6>
__device__ __forceinline__ void walk_tile(const LdsGroup &g, const LdsBatch &b, const uint8_t *lds, uint32_t wt, uint32_t lane)
{
	const uint32_t hb = g.hb, off_rec = g.off_rec;
	const uint32_t final_code = g.final_code;
	uint32_t e[C], so[C], so0[C], chain[C];
	int32_t lo[C], hi[C], keep[C];
	uint32_t *stage = b.stage + (size_t)wt * (C * kChainBytes * 64);
	// (six named register sets, picked by a branch on the wave-uniform group number: an array of them
	// indexed by the group would live in scratch)
	uint4 p0[C], p1[C], p2[C], p3[C], p4[C], p5[C];
	static_assert(kGroups == 6, "six register sets");
	const uint4 *text16 = (const uint4 *)b.text;
	const uint32_t groups = (kChainBytes + hb) >> 4;
#pragma unroll
	for (int c = 0; c < C; c++) {
		chain[c] = (wt * C + c) * 64 + lane;
		const uint32_t base = chain[c] << kLogChain;
		// walk bytes are numbered j = 0 .. S + hb - 1 from base - hb; the chain's own are j >= hb
		const uint32_t lead = min(hb, base);
		const uint32_t len = base >= g.n ? 0u : min(kChainBytes, g.n - base);
		lo[c] = (int32_t)(hb - lead);
		hi[c] = (int32_t)(hb + len);
		// records in front of drop_before belong to the shard's halo (acm_scan_shard_async)
		keep[c] = (int32_t)hb + (b.drop_before > base ? (int32_t)min(b.drop_before - base, kChainBytes) : 0);
		e[c] = base <= hb ? (b.init_ptr ? *b.init_ptr : b.init_code) : g.root_code;   // a chain this close to the start begins at byte 0, in the carried-in state
		so0[c] = so[c] = (uint32_t)c * (kChainBytes * 64) + lane;
		// all of the lane's text up front, a chain's loads back to back (scan.hip, walk_tile PRE: taken a
		// group per trip, a 64-byte line is fetched by four loads microseconds apart -- and again)
		auto fetch = [&](uint32_t gi) -> uint4 {
			const int64_t at = (int64_t)base - hb + (int64_t)gi * 16;
			if (gi < groups && (!LIVE || (at >= 0 && at < (int64_t)g.n)))
				return text16[at >> 4];
			return make_uint4(0, 0, 0, 0);
		};
		p0[c] = fetch(0);
		p1[c] = fetch(1);
		p2[c] = fetch(2);
		p3[c] = fetch(3);
		p4[c] = fetch(4);
		if constexpr (NG > 5)
			p5[c] = fetch(5);
	}
	const uint32_t hg = hb >> 4;   // groups that are halo
	// (rolled loops over the groups, a wave-uniform pick of the register set: sixteen steps of C chains
	// are the loop body -- unrolled over the groups too the kernel would not fit the instruction cache)
#define ACM_PICK(gi)                                                                     \
	{                                                                                \
		const uint32_t gu = __builtin_amdgcn_readfirstlane(gi);                  \
		_Pragma("unroll") for (int c = 0; c < C; c++)                            \
		{                                                                        \
			if (gu == 0) w[c] = p0[c];                                       \
			else if (gu == 1) w[c] = p1[c];                                  \
			else if (gu == 2) w[c] = p2[c];                                  \
			else if (gu == 3) w[c] = p3[c];                                  \
			else if (NG <= 5 || gu == 4) w[c] = p4[c];                       \
			else w[c] = p5[c];                                               \
		}                                                                        \
	}
#define ACM_STEP(K, EM) step_all<C, K, LIVE, EM>(lds, off_rec, final_code, w, e, so, stage, sb + ((uint32_t)(K) << 16), lo, hi, keep, j0 + (K))
#define ACM_STEPS(EM) ACM_STEP(0, EM); ACM_STEP(1, EM); ACM_STEP(2, EM); ACM_STEP(3, EM); ACM_STEP(4, EM); ACM_STEP(5, EM); \
	ACM_STEP(6, EM); ACM_STEP(7, EM); ACM_STEP(8, EM); ACM_STEP(9, EM); ACM_STEP(10, EM); ACM_STEP(11, EM); ACM_STEP(12, EM); \
	ACM_STEP(13, EM); ACM_STEP(14, EM); ACM_STEP(15, EM)
	const uint32_t groups_walked = groups;
#pragma nounroll
	for (uint32_t gi = 0; gi < min(hg, groups_walked); gi++) {   // the halo: only the state matters
		uint4 w[C];
		ACM_PICK(gi);
		const int32_t j0 = (int32_t)gi * 16;
		const uint32_t sb = 0;
		ACM_STEPS(false);
	}
#pragma nounroll
	for (uint32_t gi = hg; gi < groups_walked; gi++) {
		uint4 w[C];
		ACM_PICK(gi);
		const int32_t j0 = (int32_t)gi * 16;
		const uint32_t sb = ((uint32_t)j0 - hb) << 16;   // step inside the chain, in the upper half of a staged record
		ACM_STEPS(true);
	}
#undef ACM_STEPS
#undef ACM_STEP
#undef ACM_PICK
	uint32_t total = 0;
#pragma unroll
	for (int c = 0; c < C; c++) {
		const uint32_t k = (so[c] - so0[c]) >> 6;
		if (chain[c] < g.n_chains) {
			b.cnt[chain[c]] = (uint8_t)k;
			total += k;
			if (chain[c] == g.n_chains - 1)
				b.misc[0] = e[c] & 0xFFFFu;
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		total += __shfl_xor(total, o, 64);
	if (lane == 0)
		b.tile_total[wt] = total;
}

template <int C, bool ASM, int NG>
__global__ __launch_bounds__(kWalkBlock) void k_lds_walk(LdsGroup g)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	{
		uint4 *dst = (uint4 *)lds;
		const uint32_t n16 = g.image16;
		const uint32_t rot = n16 ? (blockIdx.x * 1021u) % n16 : 0u;   // (the CUs do not all ask the same L2 channel at once)
		for (uint32_t i = threadIdx.x; i < n16; i += kWalkBlock) {
			uint32_t j = i + rot;
			j = j >= n16 ? j - n16 : j;
			dst[j] = g.image[j];
		}
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = blockIdx.x * kWalkWaves + (threadIdx.x >> 6);
	const uint32_t nwaves = gridDim.x * kWalkWaves;
	const uint32_t tile_bytes = (uint32_t)C * 64u * kChainBytes;
	const uint32_t all = g.n_tiles * g.count;
	for (uint32_t t = wave; t < all; t += nwaves) {
		const uint32_t bi = t / g.n_tiles, wt = t - bi * g.n_tiles;
		const LdsBatch &b = g.b[bi];
		const uint64_t first = (uint64_t)wt * tile_bytes;
		// no shard halo in the tile (its records in front of drop_before are to be dropped: the generic walk)
		const bool whole = first >= b.drop_before;   // (the text's end inside the tile: walk_tile_fast copes)
		if constexpr (ASM && C == 2) {
			if (whole)
				walk_tile_fast<NG>(g, b, lds, wt, lane);   // (copes with the text's first chain itself)
			else
				walk_tile<C, true, NG>(g, b, lds, wt, lane);
		} else {
			if (first + tile_bytes <= g.n && first >= b.drop_before && first >= g.hb)
				walk_tile<C, false, NG>(g, b, lds, wt, lane);
			else
				walk_tile<C, true, NG>(g, b, lds, wt, lane);
		}
	}
}

// Ordered scatter: a workgroup per 1024 chains of one batch.
template <int C, int kScatterBlock>
__global__ __launch_bounds__(kScatterBlock) void k_lds_scatter(LdsGroup g)
{
	__shared__ uint32_t wtot[kScatterBlock / 64];
	__shared__ uint32_t part[kScatterBlock / 64], part_all[kScatterBlock / 64];
	const uint32_t nb = (g.n_chains + kScatterBlock - 1) / kScatterBlock;
	const uint32_t bi = blockIdx.x / nb, blk = blockIdx.x - bi * nb;
	const LdsBatch &b = g.b[bi];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t j = blk * kScatterBlock + tid;
	const uint32_t c = j < g.n_chains ? (uint32_t)b.cnt[j] : 0u;
	uint32_t inc = c;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o, 64);
		if (lane >= (uint32_t)o)
			inc += t;
	}
	if (lane == 63)
		wtot[wv] = inc;
	// the records in front of this workgroup: the totals of the tiles in front of it (a few thousand
	// L2-resident words); workgroup 0 adds up all of them for the header cell
	constexpr uint32_t kTilesPerBlock = kScatterBlock / (C * 64);
	uint32_t mine = 0, all = 0;
	const uint32_t mine_upto = blk * kTilesPerBlock, upto = blk == 0 ? g.n_tiles : mine_upto;
	for (uint32_t i = tid; i < upto; i += kScatterBlock) {
		const uint32_t v = b.tile_total[i];
		all += v;
		mine += i < mine_upto ? v : 0u;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		mine += __shfl_xor(mine, o, 64);
		all += __shfl_xor(all, o, 64);
	}
	if (lane == 0) {
		part[wv] = mine;
		part_all[wv] = all;
	}
	__syncthreads();
	uint32_t before = 0, total = 0;
	for (uint32_t w = 0; w < kScatterBlock / 64; w++) {
		before += part[w];
		total += part_all[w];
		before += w < wv ? wtot[w] : 0u;
	}
	uint32_t d = before + inc - c;
	if (c) {
		// chain j = (tile * C + slot) * 64 + lane: its list is stage[tile][slot][k][lane]
		const uint32_t tile = j / (C * 64), slot = (j >> 6) % C;
		const uint32_t *list = b.stage + (size_t)tile * (C * kChainBytes * 64) + slot * (kChainBytes * 64) + lane;
		const int32_t *outp = b.report_state ? (const int32_t *)g.cid2ref : g.out;
		const uint32_t base = j << kLogChain;
		// four records a round: their loads, then the four pattern lookups, then the stores -- a wave takes as
		// many rounds as its busiest lane has records, so a round should not be a chain of dependent loads
		for (uint32_t k = 0; k < c; k += 4) {
			uint32_t rec[4];
			int32_t pat[4];
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				rec[i] = list[min(k + i, c - 1) * 64];
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				pat[i] = outp[(rec[i] & 0xFFFFu) - g.root_code];
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				if (k + i < c && d + i + 2 < b.plane_capacity) {
					b.pat_plane[1 + d + i] = pat[i];
					b.off_plane[1 + d + i] = (int32_t)(base + (rec[i] >> 16)) + b.off_shift;
				}
			d += 4;
		}
	}
	if (blk == 0 && tid == 0) {   // header and trailer cells (compactarray.cl:49-55)
		const int32_t last_ref = (int32_t)g.cid2ref[(b.misc[0] & 0xFFFFu) - g.root_code];
		b.misc[2] = (uint32_t)ACM_SCAN_MODE_CHAIN;   // acm_scan_path_taken
		uint32_t tail = total + 1;
		if (tail > b.plane_capacity - 1)
			tail = b.plane_capacity - 1;
		b.pat_plane[0] = (int32_t)total;
		b.off_plane[0] = (int32_t)total;
		b.pat_plane[tail] = last_ref;
		b.off_plane[tail] = last_ref;
	}
}

// The same, built to run in the SHADOW of another stream's walk kernel, whose workgroup leaves a CU 128
// registers per lane and SIMD, a few hundred bytes of LDS and a fifth of the issue slots: with two waves per SIMD
// the levels of dependent loads are what it costs, so a thread takes Q chains (T apart, i.e. four tiles apart)
// through them together -- counts and tile totals; scan; up to 12 staged records of its chains in one level, their
// patterns in the next, then the stores -- and a workgroup covers T * Q chains: 256 workgroups per 32 MiB batch.
template <int C, int T, int Q>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lds_scatter_wide(LdsGroup g)   // (64 registers: two of its waves per SIMD next to a walk workgroup)
{
	constexpr uint32_t W = T / 64, kPer = T * Q, kSlots = 12;
	static_assert(Q == 4 && T % (C * 64) == 0, "four chains a thread, whole tiles per slab");
	// (s_setprio 3 here makes this kernel twice as fast next to a walk kernel and the job 9 % slower: the walk is
	// bound by instruction issue, what these waves gain its waves lose, and more)
	__shared__ uint32_t wtot[Q * W];
	__shared__ uint32_t part[W], part_all[W];
	const uint32_t nb = (g.n_chains + kPer - 1) / kPer;
	const uint32_t bi = blockIdx.x / nb, blk = blockIdx.x - bi * nb;
	const LdsBatch &b = g.b[bi];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t j0 = blk * kPer + tid;
	uint32_t c[Q], inc[Q];
#pragma unroll
	for (int q = 0; q < Q; q++) {
		const uint32_t j = j0 + q * T;
		c[q] = j < g.n_chains ? (uint32_t)b.cnt[j] : 0u;
	}
	constexpr uint32_t kTilesPer = kPer / (C * 64);
	uint32_t mine = 0, all = 0;
	const uint32_t mine_upto = blk * kTilesPer, upto = blk == 0 ? g.n_tiles : mine_upto;
	for (uint32_t i = tid; i < upto; i += T) {
		const uint32_t v = b.tile_total[i];
		all += v;
		mine += i < mine_upto ? v : 0u;
	}
#pragma unroll
	for (int q = 0; q < Q; q++)
		inc[q] = c[q];
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
		for (int q = 0; q < Q; q++) {
			const uint32_t t = __shfl_up(inc[q], o, 64);
			if (lane >= (uint32_t)o)
				inc[q] += t;
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		mine += __shfl_xor(mine, o, 64);
		all += __shfl_xor(all, o, 64);
	}
	if (lane == 63) {
#pragma unroll
		for (int q = 0; q < Q; q++)
			wtot[q * W + wv] = inc[q];
	}
	if (lane == 0) {
		part[wv] = mine;
		part_all[wv] = all;
	}
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (uint32_t w = 0; w < W; w++) {
		before += part[w];
		total += part_all[w];
	}
	// where the records of the thread's chain q go: behind the tiles in front of the workgroup, the slabs in front
	// of q and the waves in front of this one in slab q
	uint32_t d[Q];
	uint32_t run = before;
#pragma unroll
	for (int q = 0; q < Q; q++) {
#pragma unroll
		for (uint32_t w = 0; w < W; w++) {
			if (w == wv)
				d[q] = run;   // (select, not a branch: wv is uniform in the wave)
			run += wtot[q * W + w];
		}
		d[q] += inc[q] - c[q];
	}
	const uint32_t cum1 = c[0], cum2 = cum1 + c[1], cum3 = cum2 + c[2], tot = cum3 + c[3];
	// chain j = (tile * C + slot) * 64 + lane: its list is stage[tile][slot][k][lane]; chain q of the thread is
	// T chains = T / 64 slots further
	const uint32_t *list0 = b.stage + (size_t)(j0 >> 6) * (kChainBytes * 64) + lane;
	constexpr uint32_t kSlab = (T / 64) * (kChainBytes * 64);
	const int32_t *outp = b.report_state ? (const int32_t *)g.cid2ref : g.out;
	const uint32_t base0 = j0 << kLogChain;
	const uint32_t cap = b.plane_capacity;
	for (uint32_t r0 = 0; __builtin_amdgcn_ballot_w64(r0 < tot); r0 += kSlots) {
		uint32_t rec[kSlots];
		int32_t pat[kSlots];
#pragma unroll
		for (uint32_t i = 0; i < kSlots; i++) {
			const uint32_t idx = r0 + i;
			const uint32_t q = (idx >= cum1) + (idx >= cum2) + (idx >= cum3);
			const uint32_t k = idx - (q == 0 ? 0u : q == 1 ? cum1 : q == 2 ? cum2 : cum3);
			rec[i] = idx < tot ? list0[q * kSlab + k * 64] : 0u;
		}
#pragma unroll
		for (uint32_t i = 0; i < kSlots; i++)
			pat[i] = r0 + i < tot ? outp[(rec[i] & 0xFFFFu) - g.root_code] : 0;
#pragma unroll
		for (uint32_t i = 0; i < kSlots; i++) {
			const uint32_t idx = r0 + i;
			const uint32_t q = (idx >= cum1) + (idx >= cum2) + (idx >= cum3);
			const uint32_t k = idx - (q == 0 ? 0u : q == 1 ? cum1 : q == 2 ? cum2 : cum3);
			const uint32_t at = (q == 0 ? d[0] : q == 1 ? d[1] : q == 2 ? d[2] : d[3]) + k;
			if (idx < tot && at + 2 < cap) {
				b.pat_plane[1 + at] = pat[i];
				b.off_plane[1 + at] = (int32_t)(base0 + q * (T << kLogChain) + (rec[i] >> 16)) + b.off_shift;
			}
		}
	}
	if (blk == 0 && tid == 0) {   // header and trailer cells (compactarray.cl:49-55)
		const int32_t last_ref = (int32_t)g.cid2ref[(b.misc[0] & 0xFFFFu) - g.root_code];
		b.misc[2] = (uint32_t)ACM_SCAN_MODE_CHAIN;   // acm_scan_path_taken
		uint32_t tail = total + 1;
		if (tail > b.plane_capacity - 1)
			tail = b.plane_capacity - 1;
		b.pat_plane[0] = (int32_t)total;
		b.off_plane[0] = (int32_t)total;
		b.pat_plane[tail] = last_ref;
		b.off_plane[tail] = last_ref;
	}
}

template <typename T>
int to_device(T **dptr, const T *src, size_t count, size_t *total)
{
	const size_t bytes = (count ? count : 1) * sizeof(T);
	ACM_HIP_TRY(hipMalloc((void **)dptr, bytes));
	if (count)
		ACM_HIP_TRY(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
	*total += bytes;
	return ACM_OK;
}

}  // namespace

namespace acm {

uint32_t lds_walk_max_group() { return kMaxGroup; }

// Builds the LDS form of the automaton and uploads it; d->lds_ok says whether the set qualifies.
int lds_walk_prepare(const acm_automaton *a, acm_dfa *d)
{
	d->lds_ok = false;
	if (getenv("ACM_SCAN_NO_LDSWALK"))   // debugging aid: the row-in-LDS / cold-plane walk kernels of scan.hip instead
		return ACM_OK;
	const uint32_t L = (uint32_t)a->max_pattern_len;
	const uint32_t hb = ((L > 1 ? L - 1 : 0u) + 15u) & ~15u;
	if (hb > kMaxHaloGroups * 16)
		return ACM_OK;
	CompactTables t;
	build_compact(*a, t, kCompactLdsBytes);
	if (!t.ok)
		return ACM_OK;
	std::vector<int32_t> outp(t.n);
	for (uint32_t c = 0; c < t.n; c++) {
		const uint32_t r = t.cid2ref[c];
		outp[c] = a->is_final_ref(r) ? a->head_of(r) : -1;
	}
	int rc = to_device(&d->d_lds_image, t.image.data(), t.image.size(), &d->device_bytes);
	if (rc == ACM_OK) rc = to_device(&d->d_lds_out, outp.data(), outp.size(), &d->device_bytes);
	if (rc == ACM_OK) rc = to_device(&d->d_lds_cid2ref, t.cid2ref.data(), t.cid2ref.size(), &d->device_bytes);
	std::vector<uint32_t> codes(t.n);
	for (uint32_t r = 0; r < t.n; r++)
		codes[r] = t.code_of_ref(r);
	if (rc == ACM_OK) rc = to_device(&d->d_lds_ref2code, codes.data(), codes.size(), &d->device_bytes);
	if (rc != ACM_OK)
		return rc;
	d->lds_image_bytes = t.image_bytes;
	d->lds_off_rec = t.off_rec;
	d->lds_root_code = t.root_code();
	d->lds_final_code = t.final_code();
	d->lds_halo = hb;
	d->lds_rows = t.rows;
	d->lds_ref2code.resize(t.n);
	for (uint32_t r = 0; r < t.n; r++)
		d->lds_ref2code[r] = (uint16_t)t.code_of_ref(r);
	const void *kernels[] = { (const void *)k_lds_walk<kChains, true, 5>, (const void *)k_lds_walk<kChains, true, 6>,
		(const void *)k_lds_walk<kChains, false, 6> };
	for (const void *k : kernels)
		ACM_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.image_bytes));
	d->lds_ok = true;
	return ACM_OK;
}

void lds_walk_release(acm_dfa *d)
{
	hipFree(d->d_lds_image);
	hipFree(d->d_lds_out);
	hipFree(d->d_lds_cid2ref);
	hipFree(d->d_lds_ref2code);
}

// Two launches for up to 16 batches of one size on one stream.  The areas of a batch's workspace are
// handed over by the caller (scan.hip lays the workspace out).
int lds_walk_enqueue(const acm_dfa *d, const LdsJob *jobs, uint32_t count, hipStream_t s, hipEvent_t after_walk,
    hipEvent_t after_walk2)
{
	if (!d->lds_ok || count == 0 || count > kMaxGroup)
		return acm::fail(ACM_ERR_ARG, "lds_walk_enqueue: %u batches", count);
	const size_t n = jobs[0].batch->n;
	LdsGroup g;
	memset(&g, 0, sizeof(g));
	g.image = (const uint4 *)d->d_lds_image;
	g.image16 = d->lds_image_bytes / 16;
	g.off_rec = d->lds_off_rec;
	g.root_code = d->lds_root_code;
	g.final_code = d->lds_final_code;
	g.out = d->d_lds_out;
	g.cid2ref = d->d_lds_cid2ref;
	g.n = (uint32_t)n;
	g.n_chains = (uint32_t)((n + kChainBytes - 1) >> kLogChain);
	constexpr int C = kChains;
	g.n_tiles = (g.n_chains + C * 64 - 1) / (C * 64);
	g.hb = d->lds_halo;
	g.count = count;
	for (uint32_t i = 0; i < count; i++) {
		const acm_scan_batch *in = jobs[i].batch;
		if (in->n != n)
			return acm::fail(ACM_ERR_ARG, "lds_walk_enqueue: the batches of a group must have one size");
		LdsBatch &b = g.b[i];
		b.text = (const uint8_t *)in->d_text;
		b.stage = jobs[i].stage;
		b.cnt = jobs[i].cnt;
		b.tile_total = jobs[i].tile_total;
		b.misc = jobs[i].misc;
		b.pat_plane = in->d_pat_plane;
		b.off_plane = in->d_off_plane;
		b.init_code = d->lds_ref2code[(size_t)in->init_state];
		b.init_ptr = jobs[i].init_ptr;
		b.drop_before = (uint32_t)in->halo;
		b.off_shift = (int32_t)in->offset_shift;
		b.plane_capacity = (uint32_t)(in->plane_capacity > 0xFFFFFFFFul ? 0xFFFFFFFFul : in->plane_capacity);
		b.report_state = in->report == ACM_REPORT_STATE ? 1 : 0;
	}
	const uint32_t all_tiles = g.n_tiles * count;
	uint32_t blocks = (all_tiles + kWalkWaves - 1) / kWalkWaves;
	if (blocks > (uint32_t)d->num_cus)
		blocks = (uint32_t)d->num_cus;
	static const bool plain = getenv("ACM_LDS_NOASM") != nullptr;   // debugging aid: the compiler's version of the step
	static const int sblock = getenv("ACM_LDS_SCATTER_BLOCK") ? atoi(getenv("ACM_LDS_SCATTER_BLOCK")) : 0;
	if (plain)
		hipLaunchKernelGGL((k_lds_walk<C, false, 6>), dim3(blocks), dim3(kWalkBlock), d->lds_image_bytes, s, g);
	else if (g.hb <= 16)
		hipLaunchKernelGGL((k_lds_walk<C, true, 5>), dim3(blocks), dim3(kWalkBlock), d->lds_image_bytes, s, g);   // 96 registers instead of 104
	else
		hipLaunchKernelGGL((k_lds_walk<C, true, 6>), dim3(blocks), dim3(kWalkBlock), d->lds_image_bytes, s, g);
	if (after_walk)
		ACM_HIP_TRY(hipEventRecord(after_walk, s));
	if (after_walk2)
		ACM_HIP_TRY(hipEventRecord(after_walk2, s));
	auto sblocks = [&](uint32_t per) { return dim3(((g.n_chains + per - 1) / per) * count); };
	if (sblock == 1024)   // debugging aid (ACM_LDS_SCATTER_BLOCK=1024): the plain scatter, a thread per chain
		hipLaunchKernelGGL((k_lds_scatter<C, 1024>), sblocks(1024), dim3(1024), 0, s, g);
	else
		hipLaunchKernelGGL((k_lds_scatter_wide<C, 512, 4>), sblocks(2048), dim3(512), 0, s, g);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

// what a batch of n bytes needs of each workspace area (scan.hip checks them against its layout)
void lds_walk_needs(const acm_dfa *d, size_t n, size_t *stage_words, size_t *cnt_bytes, size_t *tile_words)
{
	const size_t chains = (n + kChainBytes - 1) >> kLogChain;
	constexpr int C = kChains;
	(void)d;
	const size_t tiles = (chains + C * 64 - 1) / (C * 64);
	*stage_words = tiles * C * kChainBytes * 64;
	*cnt_bytes = chains;
	*tile_words = tiles;
}

}  // namespace acm
