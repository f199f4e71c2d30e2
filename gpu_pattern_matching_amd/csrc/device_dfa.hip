// Device-resident DFA: HBM/LDS layout + upload.
//
// Replaces the d_trans half of acsm_gen_state_table (acsmx.c:618-666).  The
// reference ships one int32 [state][2][256] table (2 KiB per state, final
// transitions stored negated, pattern index in a second plane).  Here:
//
//   cold   u32 [states][256]   next state, dev numbering, 1 KiB rows.  Final
//                              states are numbered last, so "transition is
//                              final" is  next >= first_final  -- no flag
//                              bits, no second plane.
//   hot    u16 [H][256]        the first H (<= 256) non-final states in BFS
//                              order (root, depth 1, ...): copied to LDS by
//                              the scan kernel.  A cell holds the next state,
//                              or 0xFFFF when it does not fit (then the lane
//                              reads the cold plane).
//   out    i32 [states]        pattern index a final state reports (the head
//                              of its match list, acsmx.c:650), -1 otherwise
//   dev2ref u32 [states]       back to the reference's numbering (last_state)
//   depth_cum u32 [L+2]        non-final ids < depth_cum[m]  <=>  depth <= m
//   depth_final u16 [finals]   depth of final state first_final + i
//   ffinfo u32 [states], ref2dev u32 [states], in_byte u8 [states]:
//                              fast-forward along unary trie paths for the
//                              deep walks of the resolve stage (ref ids are
//                              consecutive along such a path)
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "acm_internal.h"
#include "device_dfa.h"

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
	d->max_pattern_len = (uint32_t)a->max_pattern_len;
	d->ref2dev = a->ref2dev;
	d->dev2ref_host = a->dev2ref;

	int rc = ACM_OK;
	try {
		const std::vector<uint32_t> &rows = a->dense_rows();
		uint32_t n = a->num_states;

		uint32_t H = a->first_final < acm::kHotRowsMax ? a->first_final : acm::kHotRowsMax;
		std::vector<uint16_t> hot((size_t)H * 256);
		for (size_t i = 0; i < hot.size(); i++)
			hot[i] = (uint16_t)(rows[i] < acm::kHotSentinel ? rows[i] : acm::kHotSentinel);
		d->hot_rows = H;

		std::vector<int32_t> outp(n);
		for (uint32_t s = 0; s < n; s++) {
			uint32_t r = a->dev2ref[s];
			outp[s] = a->is_final_ref(r) ? a->head_of(r) : -1;
		}
		std::vector<uint16_t> dfin(n - a->first_final);
		for (uint32_t s = a->first_final; s < n; s++)
			dfin[s - a->first_final] = a->depth[a->dev2ref[s]];

		// bigram image: cells 0..255 = root row; cell 256 + (prev | byte << 8) =
		// delta(delta(root, prev), byte), 0xFFFF when the id does not fit
		std::vector<uint16_t> t2(256 + 65536);
		for (uint32_t c = 0; c < 256; c++)
			t2[c] = (uint16_t)(rows[c] < acm::kHotSentinel ? rows[c] : acm::kHotSentinel);
		for (uint32_t prev = 0; prev < 256; prev++) {
			const uint32_t *row = &rows[(size_t)rows[prev] * 256];
			for (uint32_t c = 0; c < 256; c++)
				t2[256 + (prev | (c << 8))] =
				    (uint16_t)(row[c] < acm::kHotSentinel ? row[c] : acm::kHotSentinel);
		}
		// trigram filter: every 3-byte string that is a trie node (depth-3 state)
		std::vector<uint8_t> bloom(16384, 0);
		for (uint32_t r = 0; r < n; r++) {
			if (a->depth[r] != 3)
				continue;
			const uint32_t p2 = a->parent[r], p1 = a->parent[p2];
			const uint32_t tri = (uint32_t)a->in_byte[p1] | ((uint32_t)a->in_byte[p2] << 8) |
			    ((uint32_t)a->in_byte[r] << 16);
			const uint32_t h = (((tri * 0x9E3779u) & 0xFFFFFFFFu) >> 15) & (16384 * 8 - 1);
			bloom[h >> 3] |= (uint8_t)(1u << (h & 7));
		}
		d->cum1 = a->depth_cum.size() > 1 ? a->depth_cum[1] : a->depth_cum[0];
		d->d2lo = a->depth_cum.size() > 1 ? a->depth_cum[1] : a->first_final;
		d->d2hi = a->depth_cum.size() > 2 ? a->depth_cum[2] : d->d2lo;
		// the bigram walk pays off when most text bytes lead somewhere from the root
		// (binary signature sets); word lists keep the BFS hot rows
		uint32_t firsts = 0;
		for (uint32_t c = 0; c < 256; c++)
			firsts += rows[c] != 0;
		(void)firsts;   // measured on MI355X: the hot-row walk with 4 chains per lane is
		d->bigram_default = false;   // faster on every fixture so far; bigram stays opt-in
		d->use_bigram = d->bigram_default;

		std::vector<uint32_t> ffinfo(n);
		for (uint32_t s = 0; s < n; s++) {
			const uint32_t r = a->dev2ref[s];
			ffinfo[s] = r | ((uint32_t)a->ff_run[r] << 24);
		}
		std::vector<uint8_t> inb(a->in_byte);
		inb.resize((size_t)n + 32, 0);
		std::vector<uint8_t> ffr(a->ff_run);
		ffr.resize((size_t)n + 32, 0);

		rc = upload(&d->d_cold, rows.data(), rows.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_ffinfo, ffinfo.data(), ffinfo.size(), &d->device_bytes);
		if (rc == ACM_OK)
			rc = upload(&d->d_ref2dev, a->ref2dev.data(), a->ref2dev.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_in_byte, inb.data(), inb.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_ff_run, ffr.data(), ffr.size(), &d->device_bytes);
		if (rc == ACM_OK) {  // one image: T2 cells, then the filter bytes (LDS copy is one sweep)
			std::vector<uint16_t> image(t2);
			image.resize(t2.size() + bloom.size() / 2);
			memcpy(image.data() + t2.size(), bloom.data(), bloom.size());
			rc = upload(&d->d_t2, image.data(), image.size(), &d->device_bytes);
			d->d_bloom = (uint8_t *)(d->d_t2 + t2.size());
		}
		if (rc == ACM_OK) rc = upload(&d->d_hot, hot.data(), hot.size(), &d->device_bytes);
		if (rc == ACM_OK) rc = upload(&d->d_out, outp.data(), outp.size(), &d->device_bytes);
		if (rc == ACM_OK)
			rc = upload(&d->d_dev2ref, a->dev2ref.data(), a->dev2ref.size(), &d->device_bytes);
		if (rc == ACM_OK)
			rc = upload(&d->d_depth_cum, a->depth_cum.data(), a->depth_cum.size(),
			    &d->device_bytes);
		if (rc == ACM_OK)
			rc = upload(&d->d_depth_final, dfin.data(), dfin.size(), &d->device_bytes);
	} catch (const std::bad_alloc &) {
		rc = acm::fail(ACM_ERR_NOMEM, "acm_dfa_upload: out of host memory");
	}
	if (rc != ACM_OK) {
		acm_dfa_release(d);
		return rc;
	}
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
		hipFree(d->d_hot);
		hipFree(d->d_out);
		hipFree(d->d_dev2ref);
		hipFree(d->d_depth_cum);
		hipFree(d->d_depth_final);
		hipFree(d->d_ffinfo);
		hipFree(d->d_ref2dev);
		hipFree(d->d_in_byte);
		hipFree(d->d_ff_run);
		hipFree(d->d_t2);   // d_bloom points into the same allocation
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
