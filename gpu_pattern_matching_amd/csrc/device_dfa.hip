// Device-resident DFA: HBM/LDS layout + upload.
//
// Replaces the d_trans half of acsm_gen_state_table (acsmx.c:618-666).  The
// reference ships one int32 [state][2][256] table (2 KiB per state, final
// transitions stored negated, pattern index in a second plane).  Here, in the
// device numbering of acm_internal.h (hot rows, other non-finals in reference
// order, finals last):
//
//   cold   u32 [states][256]    1 KiB rows, next state; "the transition is
//                               final" is  next >= first_final  -- no flag
//                               bits.  The walk kernel reads only this plane.
//   deep   u64 [states][256]    next | depth(next) << 32 | run(next) << 48: what
//                               the exact walks (boundary resolve, sparse
//                               walkers) need about the state they enter
//                               (merge test, fast-forward) comes back in the
//                               same load as the state -- one load, one TLB
//                               entry per step on a latency-bound path.  Costs
//                               2 KiB per state next to the 1 KiB cold row;
//                               HBM is not what this path is short of.
//   hot    u16 [H][256]         rows of the first H (<= 256) non-final states
//                               in BFS order (root, depth 1, ...): copied to
//                               LDS by the walk kernel.  A cell holds the next
//                               state, or 0xFFFF when it does not fit or is
//                               final (the lane then reads the cold plane).
//   out    i32 [states]         pattern index a final state reports (the head
//                               of its match list, acsmx.c:650), -1 otherwise
//   dev2ref u32 [states]        back to the reference's numbering (last_state)
//   in_byte u8 [states + 224]    byte on the trie edge into each state; along
//                               a unary path the states ahead are d+1, d+2, ...
//                               so 16 expected bytes are one contiguous load
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <unordered_map>

#include "acm_internal.h"
#include "device_dfa.h"
#include "lds_walk.h"
#include "sieve_tables.h"
#include "sparse.h"

extern "C" int acm_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

namespace {

template <typename T>
int upload(T **dptr, const T *src, size_t count, size_t *total)
{
	size_t bytes = (count ? count : 1) * sizeof(T);
	ACM_HIP_TRY(hipMalloc((void **)dptr, bytes));
	if (count)
		ACM_HIP_TRY(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
	*total += bytes;
	return ACM_OK;
}

// The small tables the latency-bound lookups touch (node records, hash tables, per-state
// arrays) share ONE allocation: a lookup that lands on a page nobody touched lately pays an
// address translation on top of the access, and separate hipMallocs are separate small pages.
template <typename T>
int upload_small(acm_dfa *d, T **dptr, const T *src, size_t count)
{
	const size_t bytes = ((count ? count : 1) * sizeof(T) + 255) & ~(size_t)255;
	if (d->arena && d->arena_used + bytes <= d->arena_bytes) {
		*dptr = (T *)((char *)d->arena + d->arena_used);
		d->arena_used += bytes;
		if (count)
			ACM_HIP_TRY(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
		return ACM_OK;
	}
	return upload(dptr, src, count, &d->device_bytes);
}

void free_small(acm_dfa *d, void *p)
{
	if (p && !(d->arena && (char *)p >= (char *)d->arena && (char *)p < (char *)d->arena + d->arena_bytes))
		hipFree(p);
}

// Tables of the sparse pipeline (sieve_tables.h).  sparse_ok stays false for sets the
// pipeline does not apply to (a pattern shorter than 3 bytes, or none at all).
int build_sieve(const acm_automaton &a, acm_dfa *d)
{
	d->sparse_ok = false;
	size_t shortest = SIZE_MAX;
	for (const auto &p : a.patterns)
		shortest = std::min(shortest, p.bytes.size());
	if (a.patterns.empty() || shortest < 3)
		return ACM_OK;
	const uint32_t n = a.num_states, F = a.first_final;
	uint32_t W = acm::sieve_stride((uint32_t)std::min<size_t>(shortest, 64));
	if (const char *e = getenv("ACM_SIEVE_STRIDE")) {   // debugging aid: a smaller stride than the set allows
		const uint32_t v = (uint32_t)atoi(e);
		if ((v == 1 || v == 2 || v == 4 || v == 8) && v <= W)
			W = v;
	}
	const uint32_t D = (uint32_t)std::min<size_t>(shortest, acm::kSieveMaxPrefix);
	d->sv_stride = W;
	d->sv_prefix_len = D;
	// runs of one byte that can start a pattern: D copies of b down the trie
	memset(d->sv_run_ok, 0, sizeof(d->sv_run_ok));
	for (uint32_t b = 0; b < 256; b++) {
		uint32_t s = 0, k = 0;
		for (; k < D; k++) {
			uint32_t next = UINT32_MAX;
			for (uint32_t e = a.child_begin[s]; e < a.child_begin[s + 1]; e++)
				if (a.child_list[e].byte == b)
					next = a.child_list[e].to;
			if (next == UINT32_MAX)
				break;
			s = next;
		}
		if (k == D)
			d->sv_run_ok[b >> 5] |= 1u << (b & 31);
	}

	// 3-grams at offsets < W of every pattern, with the offsets they occur at
	std::unordered_map<uint32_t, uint32_t> grams;
	grams.reserve(a.patterns.size() * W * 2);
	for (const auto &p : a.patterns)
		for (uint32_t o = 0; o < W; o++) {
			const uint32_t g = (uint32_t)p.bytes[o] | ((uint32_t)p.bytes[o + 1] << 8) | ((uint32_t)p.bytes[o + 2] << 16);
			grams[g] |= 1u << o;
		}
	// the filter's keys: the 3-grams, or the 6 bytes at the sampled offsets where every pattern has them
	const uint32_t LG = (W + 5 <= shortest && W >= 4) ? 6u : 3u;
	d->sv_gram_len = LG;
	std::unordered_map<uint64_t, bool> fkeys;
	for (const auto &p : a.patterns)
		for (uint32_t o = 0; o < W; o++) {
			const uint64_t g3 = (uint32_t)p.bytes[o] | ((uint32_t)p.bytes[o + 1] << 8) | ((uint32_t)p.bytes[o + 2] << 16);
			const uint64_t m3 = LG == 6 ? (uint32_t)p.bytes[o + 3] | ((uint32_t)p.bytes[o + 4] << 8) | ((uint32_t)p.bytes[o + 5] << 16) : 0u;
			fkeys[g3 | (m3 << 24)] = true;
		}
	uint32_t lw = acm::kSieveMinLogWords;
	while (lw < acm::kSieveMaxLogWords && ((size_t)1 << lw) < fkeys.size())
		lw++;
	if (const char *e = getenv("ACM_BLOOM_LOG_WORDS")) {   // debugging aid
		const int v = atoi(e);
		if (v >= (int)acm::kSieveMinLogWords && v <= (int)acm::kSieveMaxLogWords)
			lw = (uint32_t)v;
	}
	d->sv_bloom_log_words = lw;
	std::vector<uint32_t> bloom((size_t)1 << lw, 0);
	for (const auto &kv : fkeys) {
		const uint32_t g3 = (uint32_t)(kv.first & 0xFFFFFFu), m3 = (uint32_t)(kv.first >> 24);
		const uint32_t blk = acm::sieve_bloom_block(g3, m3, lw);
		const uint64_t bits = acm::sieve_bloom_bits(g3, m3);
		bloom[2 * blk] |= (uint32_t)bits;
		bloom[2 * blk + 1] |= (uint32_t)(bits >> 32);
	}

	// gram table: buckets of four, one gram per bucket on average (a full bucket costs the
	// lookup a second, dependent load: 2 % of the buckets)
	uint32_t lb = 4;
	while (((size_t)1 << lb) < grams.size())
		lb++;
	std::vector<uint32_t> gt((size_t)4 << lb, 0);
	uint32_t gprobes = 1;
	for (const auto &kv : grams) {
		uint32_t b = acm::sieve_gram_bucket(kv.first, lb), probes = 1;
		for (;; b = (b + 1) & ((1u << lb) - 1), probes++) {
			uint32_t *slot = &gt[(size_t)b * 4];
			int k = 0;
			while (k < 4 && slot[k] != 0)
				k++;
			if (k < 4) {
				slot[k] = kv.first | (kv.second << 24);
				break;
			}
		}
		gprobes = std::max(gprobes, probes);
	}
	d->sv_gram_log_buckets = lb;
	d->sv_gram_probes = gprobes;

	// prefix table: every depth-D node under its D path bytes
	std::vector<uint32_t> nodes;
	for (uint32_t r = 0; r < n; r++)
		if (a.depth[r] == D)
			nodes.push_back(r);
	uint32_t ls = 4;   // an eighth full: a lookup ends at the first slot it reads, nearly always
	while (((size_t)1 << ls) < 8 * nodes.size())
		ls++;
	std::vector<uint32_t> pt((size_t)4 << ls, 0);
	uint32_t pprobes = 1;
	for (uint32_t r : nodes) {
		uint8_t key[12] = { 0 };
		uint32_t s = r;
		for (uint32_t k = D; k-- > 0; s = a.parent[s])
			key[k] = a.in_byte[s];
		uint32_t k0, k1, k2;
		memcpy(&k0, key, 4);
		memcpy(&k1, key + 4, 4);
		memcpy(&k2, key + 8, 4);
		const uint32_t dev = a.ref2dev[r];
		uint32_t at = acm::sieve_prefix_slot(k0, k1, k2, ls), probes = 1;
		while (pt[(size_t)at * 4 + 3] != 0) {
			at = (at + 1) & ((1u << ls) - 1);
			probes++;
		}
		pt[(size_t)at * 4 + 0] = k0;
		pt[(size_t)at * 4 + 1] = k1;
		pt[(size_t)at * 4 + 2] = k2 | ((uint32_t)a.dev_run[dev] << 16);
		pt[(size_t)at * 4 + 3] = dev;
		pprobes = std::max(pprobes, probes);
	}
	d->sv_prefix_log_slots = ls;
	d->sv_prefix_probes = pprobes;

	// node records and edges
	std::vector<acm::SieveRec> rec(n, acm::sieve_rec(0, 0, 0, false, 0, 0, 0)), edges;
	for (uint32_t dev = 0; dev < n; dev++) {
		const uint32_t r = a.dev2ref[dev];
		const uint32_t cb = a.child_begin[r], ce = a.child_begin[r + 1], nc = ce - cb;
		auto leaf = [&](uint32_t child_ref) { return a.child_begin[child_ref + 1] == a.child_begin[child_ref]; };
		auto outp = [&](uint32_t child_ref) { return a.is_final_ref(child_ref) ? (uint32_t)a.head_of(child_ref) : 0xFFFFFFFFu; };
		if (nc == 1) {
			const uint32_t c = a.child_list[cb].to, cd = a.ref2dev[c];
			rec[dev] = acm::sieve_rec(cd, a.child_list[cb].byte, 1, leaf(c), a.dev_run[cd], outp(c), c);
		} else if (nc >= 2) {
			rec[dev] = acm::sieve_rec((uint32_t)edges.size(), 0, nc, false, 0, 0, 0);
			for (uint32_t e = cb; e < ce; e++) {
				const uint32_t c = a.child_list[e].to, cd = a.ref2dev[c];
				edges.push_back(acm::sieve_edge(a.child_list[e].byte, cd, leaf(c), a.dev_run[cd], outp(c), c));
			}
		}
	}
	if (edges.size() >= (1u << 24))
		return ACM_OK;   // edge index does not fit a record: the set stays on the chain pipeline
	edges.push_back(acm::sieve_edge(0, 0, false, 0, 0, 0));
	(void)F;

	int rc = upload_small(d, &d->d_sv_bloom, bloom.data(), bloom.size());
	if (rc == ACM_OK) rc = upload_small(d, &d->d_sv_gram, gt.data(), gt.size());
	if (rc == ACM_OK) rc = upload_small(d, &d->d_sv_prefix, pt.data(), pt.size());
	acm::SieveRec *drec = nullptr, *dedges = nullptr;
	if (rc == ACM_OK) rc = upload_small(d, &drec, rec.data(), rec.size());
	if (rc == ACM_OK) rc = upload_small(d, &dedges, edges.data(), edges.size());
	d->d_sv_rec = drec;
	d->d_sv_edges = dedges;
	if (rc == ACM_OK)
		d->sparse_ok = true;
	return rc;
}

}  // namespace

extern "C" int acm_dfa_upload(const acm_automaton *a, int device, acm_dfa **out)
{
	if (!a || !a->compiled || !out)
		return acm::fail(ACM_ERR_ARG, "acm_dfa_upload: automaton not compiled");
	int ndev = acm_device_count();
	if (ndev <= 0)
		return acm::fail(ACM_ERR_NODEV, "acm_dfa_upload: no HIP device visible");
	if (device < 0 || device >= ndev)
		return acm::fail(ACM_ERR_NODEV, "acm_dfa_upload: device %d not in [0,%d)", device, ndev);
	ACM_HIP_TRY(hipSetDevice(device));

	acm_dfa *d = new (std::nothrow) acm_dfa();
	if (!d)
		return acm::fail(ACM_ERR_NOMEM, "acm_dfa_upload: out of host memory");
	d->device = device;
	d->num_states = a->num_states;
	d->first_final = a->first_final;
	d->hot_rows = a->hot_count;
	d->hot_depth1 = a->hot_depth1;
	d->max_pattern_len = (uint32_t)a->max_pattern_len;
	d->ref2dev = a->ref2dev;

	int rc = ACM_OK;
	{
		const size_t want = (((size_t)a->num_states * 48 + (8u << 20)) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
		if (hipMalloc(&d->arena, want) == hipSuccess) {
			d->arena_bytes = want;
			d->device_bytes += want;
		} else {
			d->arena = nullptr;
			(void)hipGetLastError();
		}
	}
	try {
		const std::vector<uint64_t> &rows = a->dense_rows();
		const uint32_t n = a->num_states, H = a->hot_count, F = a->first_final;

		// the planes of the chain pipeline, one column per byte class (automaton: byte_classes)
		const uint32_t ls = a->log_stride, stride = 1u << ls;
		d->log_stride = ls;
		std::vector<uint64_t> packed;
		if (ls != 8) {
			packed.assign((size_t)n << ls, 0);
			for (uint32_t s = 0; s < n; s++)
				for (uint32_t c = 0; c < stride; c++)
					packed[((size_t)s << ls) | c] = rows[((size_t)s << 8) | a->class_byte[c]];
		}
		const std::vector<uint64_t> &cells = ls != 8 ? packed : rows;
		std::vector<uint16_t> hot((((size_t)H << ls) + 7) & ~(size_t)7, (uint16_t)acm::kHotSentinel);   // whole uint4s
		for (size_t i = 0; i < ((size_t)H << ls); i++) {
			const uint32_t t = (uint32_t)cells[i];
			hot[i] = (uint16_t)((t < acm::kHotSentinel && t < F) ? t : acm::kHotSentinel);
		}
		std::vector<int32_t> outp(n);
		std::vector<uint8_t> inb((size_t)n + 224, 0);   // (the followers read up to 192 bytes past a state's own)
		for (uint32_t s = 0; s < n; s++) {
			const uint32_t r = a->dev2ref[s];
			outp[s] = a->is_final_ref(r) ? a->head_of(r) : -1;
			inb[s] = a->in_byte[r];
		}
		{
			std::vector<uint32_t> plane(cells.size());
			for (size_t i = 0; i < cells.size(); i++)
				plane[i] = (uint32_t)cells[i];
			rc = upload(&d->d_cold, plane.data(), plane.size(), &d->device_bytes);
			if (rc == ACM_OK)
				rc = upload(&d->d_deep, cells.data(), cells.size(), &d->device_bytes);
			if (rc == ACM_OK)
				rc = upload_small(d, &d->d_class, (const uint8_t *)a->byte_class, (size_t)256);
		}
		if (rc == ACM_OK) rc = upload(&d->d_hot, hot.data(), hot.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload_small(d, &d->d_out, outp.data(), outp.size());
		if (rc == ACM_OK)
			rc = upload_small(d, &d->d_dev2ref, a->dev2ref.data(), a->dev2ref.size());
		if (rc == ACM_OK) rc = upload_small(d, &d->d_in_byte, inb.data(), inb.size());
		if (rc == ACM_OK) rc = upload_small(d, &d->d_ref2dev, a->ref2dev.data(), a->ref2dev.size());

		if (rc == ACM_OK) {
			std::vector<uint16_t> dep(n);
			for (uint32_t s = 0; s < n; s++)
				dep[s] = a->depth[a->dev2ref[s]];
			rc = upload_small(d, &d->d_depth, dep.data(), dep.size());
		}
		if (rc == ACM_OK)
			rc = build_sieve(*a, d);
		if (rc == ACM_OK)
			rc = acm::lds_walk_prepare(a, d);
		// match lists, for all-patterns reporting (post.hip, acm_expand_matches_async)
		std::vector<uint32_t> lbegin(n, 0), llen(n, 0);
		for (uint32_t r = 0; r < n; r++)
			if (a->is_final_ref(r)) {
				lbegin[r] = (uint32_t)a->list_begin[r];
				llen[r] = (uint32_t)a->list_len[r];
			}
		if (rc == ACM_OK) rc = upload(&d->d_list_begin, lbegin.data(), lbegin.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_list_len, llen.data(), llen.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_list_pool, a->list_pool.data(), a->list_pool.size(), &d->device_bytes);
	} catch (const std::bad_alloc &) {
		rc = acm::fail(ACM_ERR_NOMEM, "acm_dfa_upload: out of host memory");
	}
	if (rc != ACM_OK) {
		acm_dfa_release(d);
		return rc;
	}
	if (acm::scan_prepare(d) != ACM_OK || (d->sparse_ok && acm::sparse_prepare(d) != ACM_OK)) {
		acm_dfa_release(d);
		return ACM_ERR_HIP;
	}
	if (d->sparse_ok && hipHostMalloc((void **)&d->h_giveups, 64, hipHostMallocMapped) == hipSuccess) {
		memset(d->h_giveups, 0, 64);
		if (hipHostGetDevicePointer((void **)&d->d_giveups, d->h_giveups, 0) != hipSuccess)
			d->d_giveups = nullptr;
	}
	if (!d->d_giveups && d->h_giveups) {   // no adaptive mode without the counter
		hipHostFree(d->h_giveups);
		d->h_giveups = nullptr;
	}
	(void)hipGetLastError();
	if (const char *m = getenv("ACM_SCAN_MODE")) {   // debugging aid: same as acm_scan_set_mode
		if (!strcmp(m, "chain")) d->scan_mode = ACM_SCAN_MODE_CHAIN;
		else if (!strcmp(m, "sparse")) d->scan_mode = ACM_SCAN_MODE_SPARSE;
	}
	if (const char *h = getenv("ACM_SCAN_HALO"))   // debugging aid: 0 = always speculate and resolve
		d->use_halo = atoi(h) != 0;
	if (getenv("ACM_SCAN_NO_PRELOAD"))               // debugging aid: halo mode without the up-front text loads
		d->use_preload = false;
	if (const char *g = getenv("ACM_SCAN_GRAPHS"))
		d->use_graphs = atoi(g) != 0;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess)
		d->num_cus = prop.multiProcessorCount;
	*out = d;
	return ACM_OK;
}

extern "C" void acm_dfa_release(acm_dfa *d)
{
	if (!d)
		return;
	if (hipSetDevice(d->device) == hipSuccess) {
		hipFree(d->d_cold);
		hipFree(d->d_deep);
		hipFree(d->d_hot);
		free_small(d, d->d_out);
		free_small(d, d->d_dev2ref);
		free_small(d, d->d_in_byte);
		free_small(d, d->d_ref2dev);
		hipFree(d->d_list_begin);
		hipFree(d->d_list_len);
		hipFree(d->d_list_pool);
		free_small(d, d->d_depth);
		free_small(d, d->d_class);
		free_small(d, d->d_sv_bloom);
		free_small(d, d->d_sv_gram);
		free_small(d, d->d_sv_prefix);
		free_small(d, d->d_sv_rec);
		free_small(d, d->d_sv_edges);
		hipFree(d->arena);
		acm::lds_walk_release(d);
		if (d->h_giveups)
			hipHostFree(d->h_giveups);
		(void)hipDeviceSynchronize();   // no exec is destroyed under a launch
		for (auto &g : d->graphs)
			if (g.exec)
				hipGraphExecDestroy((hipGraphExec_t)g.exec);
		for (void *e : d->parked_graphs)
			hipGraphExecDestroy((hipGraphExec_t)e);
		for (void *e : d->profile_events)
			hipEventDestroy((hipEvent_t)e);
		for (void *e : d->profile_pool)
			hipEventDestroy((hipEvent_t)e);
	}
	delete d;
}

extern "C" size_t acm_dfa_device_bytes(const acm_dfa *d) { return d ? d->device_bytes : 0; }
extern "C" int acm_dfa_hot_rows(const acm_dfa *d) { return d ? (int)d->hot_rows : 0; }
extern "C" int acm_dfa_device(const acm_dfa *d) { return d ? d->device : -1; }
