// Construction of the LDS-resident automaton (compact_tables.h).  Host code only.
//
// The DFA restated: acsmx.c:444-486 (convert_NFA_to_DFA) fills every missing transition of a state
// from its fail state's row.  Read the other way round, row(s) = row(fail(s)) with the trie children
// of s written over it -- which is what the records below keep instead of the rows.
#include "compact_tables.h"

#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>

#include "acm_internal.h"

namespace acm {

namespace {

struct Work {
	const acm_automaton &a;
	uint32_t n, nc;
	std::vector<uint32_t> N;          // [ref][class] -> ref: the DFA over byte classes
	std::vector<uint8_t> is_row;      // [ref]
	std::vector<uint32_t> h;          // [ref] nearest state with a full row on the fail chain (itself if it has one)

	explicit Work(const acm_automaton &aut) : a(aut), n(aut.num_states), nc(aut.num_classes) {}

	const uint32_t *row(uint32_t s) const { return &N[(size_t)s * nc]; }

	void dense_over_classes()
	{
		N.assign((size_t)n * nc, 0);
		for (uint32_t s : a.bfs_order) {
			uint32_t *r = &N[(size_t)s * nc];
			if (s != 0)
				memcpy(r, row(a.fail[s]), nc * sizeof(uint32_t));
			for (uint32_t e = a.child_begin[s]; e < a.child_begin[s + 1]; e++)
				r[a.byte_class[a.child_list[e].byte]] = a.child_list[e].to;
		}
	}

	// classes on which s differs from d
	uint32_t diff(uint32_t s, uint32_t d, uint32_t *cls, uint32_t cap) const
	{
		const uint32_t *x = row(s), *y = row(d);
		uint32_t k = 0;
		for (uint32_t c = 0; c < nc; c++)
			if (x[c] != y[c]) {
				if (k < cap)
					cls[k] = c;
				k++;
			}
		return k;
	}

	// Which states keep a full row: the root, every state that neither its nearest row nor its fail
	// state explains up to two overrides, and -- shallow first -- up to `optional` of the states that
	// would need the side table (the slow path of the walk).  Returns the number of rows.
	uint32_t choose_rows(uint32_t optional)
	{
		is_row.assign(n, 0);
		h.assign(n, 0);
		uint32_t rows = 0, used = 0;
		for (uint32_t s : a.bfs_order) {
			bool full = s == 0;
			if (!full) {
				const uint32_t f = a.fail[s], hA = h[f];
				uint32_t tmp[3];
				const uint32_t kA = diff(s, hA, tmp, 3);
				const uint32_t kB = is_row[f] ? kA : diff(s, f, tmp, 3);
				if (kA > 2 && kB > 2)
					full = true;
				else if (kA > 2 && used < optional) {   // would defer to its fail state's record: worth a row while there is room
					full = true;
					used++;
				}
				if (!full)
					h[s] = hA;
			}
			if (full) {
				is_row[s] = 1;
				h[s] = s;
				rows++;
			}
		}
		return rows;
	}
};

void put16(std::vector<uint8_t> &img, size_t at, uint32_t v)
{
	img[at] = (uint8_t)v;
	img[at + 1] = (uint8_t)(v >> 8);
}
void put32(std::vector<uint8_t> &img, size_t at, uint32_t v)
{
	put16(img, at, v & 0xFFFFu);
	put16(img, at + 2, v >> 16);
}
uint32_t get16(const std::vector<uint8_t> &img, size_t at) { return (uint32_t)img[at] | ((uint32_t)img[at + 1] << 8); }
uint32_t get32(const std::vector<uint8_t> &img, size_t at) { return get16(img, at) | (get16(img, at + 2) << 16); }

// numbering + records for the rows chosen in w; false: something does not fit its field
bool emit(const Work &w, CompactTables &t, uint32_t lds_bytes)
{
	const acm_automaton &a = w.a;
	const uint32_t n = w.n, nc = w.nc;
	t.nc = nc;
	t.n = n;
	t.ref2cid.assign(n, UINT32_MAX);
	t.cid2ref.assign(n, 0);
	// Compact ids: breadth-first order -- the non-final states first, then the final ones ("final" is a compare of
	// the code with one constant).  Breadth-first, because a record's LDS banks are its id modulo 32 and the walk
	// is bound by the LDS as much as by its instructions: the lanes of a wave are mostly in shallow states, and
	// numbered this way the root, its children and their children -- the states most lanes are in -- lie side by
	// side, i.e. in different banks (in trie preorder, where a state's id depends on the sizes of the subtrees in
	// front of it, the kernel took a third longer).  A row is found through its state's record, never
	// through the id, so rows are numbered on their own, shallow first.
	std::vector<uint32_t> row_of(n, UINT32_MAX);
	uint32_t nrows = 0;
	for (uint32_t s : a.bfs_order)
		if (w.is_row[s])
			row_of[s] = nrows++;
	t.rows = nrows;
	{
		// (One more constraint: a record that defers names its fail state by id in the field the walk adds the
		// byte's class to for the row read -- every lane issues that read, needed or not, and a misaligned
		// ds_read_u16 costs the whole wave instruction extra LDS cycles (+77 % LDS time measured when a quarter of
		// the wave instructions had such a lane).  So states that are deferred TO get even ids: within each of the
		// two classes they take the even places, the others the odd ones, as long as both kinds last.)
		std::vector<uint8_t> target(n, 0);
		for (uint32_t s = 0; s < n; s++)
			if (!w.is_row[s]) {
				uint32_t tmp[3];
				if (w.diff(s, w.h[s], tmp, 3) > 2)
					target[a.fail[s]] = 1;
			}
		const std::vector<uint32_t> &order = a.bfs_order;
		if (order.size() != n || target[0] || a.is_final_ref(0))
			return false;   // (the root is id 0 and not final: what the walk starts in and what the scatter decodes codes by)
		uint32_t next = 0;
		for (int pass = 0; pass < 2; pass++) {
			if (pass == 1)
				t.first_final = next;
			std::vector<uint32_t> even, odd;   // deferred-to states, the others (breadth-first each)
			for (uint32_t s : order)
				if ((a.is_final_ref(s) ? 1 : 0) == pass)
					(target[s] ? even : odd).push_back(s);
			size_t ie = 0, io = 0;
			if (pass == 0 && !odd.empty() && odd[0] == 0) {   // the root is id 0
				t.ref2cid[0] = next;
				t.cid2ref[next++] = 0;
				io = 1;
			}
			while (ie < even.size() || io < odd.size()) {
				uint32_t s;
				if ((next & 1u) == 0 && ie < even.size())
					s = even[ie++];
				else if (io < odd.size())
					s = odd[io++];
				else
					s = even[ie++];   // only deferred-to states left: some get odd ids (slower reads, nothing else)
				t.ref2cid[s] = next;
				t.cid2ref[next++] = s;
			}
		}
		if (next != n)
			return false;
	}
	t.is_final.assign(n, 0);
	for (uint32_t s = 0; s < n; s++)
		t.is_final[t.ref2cid[s]] = a.is_final_ref(s) ? 1 : 0;
	// the layout first: a code is the address of a record.  The class map (a compile-time LDS address for the
	// walk), the rows behind it (a row is named by the byte address of its first cell), then the records
	auto up = [](uint32_t v, uint32_t al) { return (v + al - 1) / al * al; };
	t.off_cls = 0;
	t.off_rows = 256;
	t.off_rec = up(t.off_rows + t.rows * nc * 2, 16);
	t.off_side = t.off_rec + n * 8;
	t.image_bytes = up(t.off_side, 16);
	if (t.off_rows + (size_t)t.rows * nc * 2 > kCompactSideBase || nc > 127)
		return false;
	if (t.image_bytes > lds_bytes)
		return false;
	auto code = [&](uint32_t ref) { return t.code_of_ref(ref); };

	std::vector<uint64_t> rec(n, 0);
	t.simple = t.side_row = t.side_link = 0;
	for (uint32_t s = 0; s < n; s++) {
		const uint32_t cid = t.ref2cid[s];
		uint32_t cls[3] = { 0, 0, 0 };
		uint32_t k = 0, next16;
		if (w.is_row[s]) {
			next16 = t.off_rows + row_of[s] * nc * 2;
		} else {
			const uint32_t f = a.fail[s], hA = w.h[s];
			k = w.diff(s, hA, cls, 3);
			next16 = t.off_rows + row_of[hA] * nc * 2;
			if (k > 2) {   // two overrides of its own on top of what the fail state's record says
				k = w.diff(s, f, cls, 3);
				next16 = kCompactSideBase + t.ref2cid[f];
				if (k > 2 || w.is_row[f])
					return false;
				t.side_link++;
			} else if (k == 0) {
				t.simple++;
			} else {
				t.side_row++;
			}
		}
		const uint32_t c1 = k > 0 ? 2 * cls[0] : kCompactNoClass, t1 = k > 0 ? code(w.row(s)[cls[0]]) : 0u;
		const uint32_t c2 = k > 1 ? 2 * cls[1] : kCompactNoClass, t2 = k > 1 ? code(w.row(s)[cls[1]]) : 0u;
		rec[cid] = (uint64_t)(t1 | (c1 << 16) | (c2 << 24)) | ((uint64_t)(t2 | (next16 << 16)) << 32);
	}
	t.nside = 0;
	t.image.assign(t.image_bytes, 0);
	for (uint32_t s = 0; s < n; s++) {
		if (row_of[s] == UINT32_MAX)
			continue;
		const uint32_t *src = w.row(s);
		for (uint32_t c = 0; c < nc; c++)
			put16(t.image, t.off_rows + ((size_t)row_of[s] * nc + c) * 2, code(src[c]));
	}
	for (uint32_t c = 0; c < n; c++) {
		put32(t.image, t.off_rec + (size_t)c * 8, (uint32_t)rec[c]);
		put32(t.image, t.off_rec + (size_t)c * 8 + 4, (uint32_t)(rec[c] >> 32));
	}
	for (uint32_t b = 0; b < 256; b++)
		t.image[t.off_cls + b] = (uint8_t)(2 * a.byte_class[b]);
	return true;
}

}  // namespace

void build_compact(const acm_automaton &a, CompactTables &t, uint32_t lds_bytes)
{
	t = CompactTables();
	if (!a.compiled || a.num_states == 0 || a.num_states > kCompactMaxStates || a.log_stride == 8 || a.num_classes > 120)
		return;
	Work w(a);
	w.dense_over_classes();
	// The rows that must be, then -- shallow states first -- as many of the states that would need the
	// side table (the walk's slow path) as the LDS has room for.  A row costs nc cells, but takes the
	// state's side entry away and often those of the states that fall back to it: the size is not
	// monotonic in the number of rows, so the largest number that fits is found by bisection over
	// "fits", and the result is whatever the last fitting attempt produced.
	const uint32_t must = w.choose_rows(0);
	CompactTables best;
	if (!emit(w, best, lds_bytes))
		return;   // even with only the rows that must be full it does not fit (or a field overflows)
	uint32_t best_rows = must;
	uint32_t lo = 0, hi = w.n;   // lo: fits; hi: does not (or is more than there are candidates)
	while (hi - lo > 1) {
		const uint32_t mid = lo + (hi - lo) / 2;
		const uint32_t rows = w.choose_rows(mid);
		CompactTables cand;
		if (emit(w, cand, lds_bytes)) {
			lo = mid;
			if (rows >= best_rows) {
				best = std::move(cand);
				best_rows = rows;
			}
		} else {
			hi = mid;
		}
	}
	t = std::move(best);
	t.promoted = best_rows - must;
	t.ok = true;
}

uint32_t compact_step(const CompactTables &t, uint32_t e, uint8_t byte, uint32_t *hops)
{
	const std::vector<uint8_t> &m = t.image;
	const uint32_t c2 = m[t.off_cls + byte];   // 2 * class
	size_t at = (size_t)(e & 0xFFFFu) << 3;    // the code is the record's address / 8
	for (uint32_t links = 0;; links++) {
		const uint32_t rx = get32(m, at), ry = get32(m, at + 4);
		if (hops)
			*hops = links;
		if (((rx >> 16) & 0xFFu) == c2)
			return rx & 0xFFFFu;
		if ((rx >> 24) == c2)
			return ry & 0xFFFFu;
		const uint32_t nx = ry >> 16;
		if (nx < kCompactSideBase)
			return get16(m, nx + c2);
		at = (size_t)t.off_rec + (size_t)(nx - kCompactSideBase) * 8;
	}
}

}  // namespace acm

// Debugging / test entry point (no device needed): builds the LDS form of a compiled automaton and
// checks EVERY (state, byte) transition of it against the dense DFA.  Returns 1 and the statistics
// when the set qualifies and every transition agrees, 0 when it does not qualify, -1 on a mismatch.
// stats: [0] states [1] classes [2] rows [3] side entries [4] image bytes [5] promoted rows
//        [6] simple records [7] side entries that end in a row [8] that defer to the fail state's record
extern "C" int acm_compact_selftest(const acm_automaton *a, uint32_t lds_bytes, uint32_t *stats)
{
	if (!a || !a->compiled)
		return acm::fail(ACM_ERR_ARG, "acm_compact_selftest: automaton not compiled");
	acm::CompactTables t;
	acm::build_compact(*a, t, lds_bytes ? lds_bytes : acm::kCompactLdsBytes);
	if (stats) {
		const uint32_t v[9] = { t.n, t.nc, t.rows, t.nside, t.image_bytes, t.promoted, t.simple, t.side_row, t.side_link };
		memcpy(stats, v, sizeof(v));
	}
	if (!t.ok)
		return 0;
	const std::vector<uint64_t> &dense = a->dense_rows();
	for (uint32_t ref = 0; ref < a->num_states; ref++) {
		const uint32_t e = t.code_of_ref(ref);
		const uint64_t *row = &dense[(size_t)a->ref2dev[ref] * 256];
		for (uint32_t b = 0; b < 256; b++) {
			const uint32_t want = t.code_of_ref(a->dev2ref[(uint32_t)row[b]]);
			const uint32_t got = acm::compact_step(t, e, (uint8_t)b, nullptr);
			if (got != want) {
				acm::fail(ACM_ERR_LIMIT, "compact tables: state %u byte %u -> code %u, DFA says %u", ref, b, got, want);
				return -1;
			}
		}
	}
	return 1;
}

// Debugging / tuning aid (no device needed): walks `text` through the LDS form from the root and
// counts how the steps were decided: [0] steps, [1] by the record or a row alone (the kernel's
// straight-line path), [2] by one side entry, [3] by more than one hop, [4] final states entered.
extern "C" int acm_compact_profile(const acm_automaton *a, const unsigned char *text, size_t n, uint64_t *counts)
{
	if (!a || !a->compiled || !counts)
		return acm::fail(ACM_ERR_ARG, "acm_compact_profile: bad arguments");
	acm::CompactTables t;
	acm::build_compact(*a, t, acm::kCompactLdsBytes);
	memset(counts, 0, 5 * sizeof(uint64_t));
	if (!t.ok)
		return 0;
	uint32_t e = t.root_code();
	for (size_t i = 0; i < n; i++) {
		uint32_t hops = 0;
		const uint32_t from = e;
		e = acm::compact_step(t, e, text[i], &hops);
		if (hops && getenv("ACM_COMPACT_DEBUG")) {
			static uint64_t hist[24][4];
			static uint64_t seen = 0;
			const uint32_t ref = t.cid2ref[t.cid_of_code(from)];
			const uint32_t nch = a->child_begin[ref + 1] - a->child_begin[ref];
			hist[std::min<uint32_t>(a->depth[ref], 23)][std::min<uint32_t>(nch, 3)]++;
			if (++seen == 20000) {
				for (int d = 0; d < 24; d++)
					if (hist[d][0] + hist[d][1] + hist[d][2] + hist[d][3])
						fprintf(stderr, "[compact] deferring visits at depth %2d: children 0:%llu 1:%llu 2:%llu 3+:%llu\n", d,
						    (unsigned long long)hist[d][0], (unsigned long long)hist[d][1], (unsigned long long)hist[d][2], (unsigned long long)hist[d][3]);
			}
		}
		counts[0]++;
		counts[hops == 0 ? 1 : hops == 1 ? 2 : 3]++;
		counts[4] += e >= t.final_code() ? 1 : 0;
	}
	return 1;
}
