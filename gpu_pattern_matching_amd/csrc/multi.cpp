// Multi-GPU host logic behind the C ABI: how a text is cut into per-rank shards
// and how the ranks' compact match planes get to one of them.
//
// The reference is single-device (ocl_aho_grep.c:498-502 hands the same -D to
// every worker).  Here the text shards by range: rank g scans
// [g*N/G, (g+1)*N/G) plus the max_pattern_len - 1 bytes in front of it (the halo),
// from state 0, and drops records that end inside the halo
// (acm_scan_shard_async): identical to the serial scan, because the DFA state
// depends only on the last max_pattern_len bytes.  The DFA is replicated; the
// one exchange is a fixed-capacity gather of the planes over RCCL -- point to
// point, every rank straight to the root over its own xGMI link.  RCCL is
// loaded on first use (dlopen): a single-GPU host never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <mutex>

#include "acm_internal.h"

extern "C" int acm_shard_plan_for(size_t n, int world, int rank, int max_pattern_len, acm_shard_plan *out)
{
	if (!out || world <= 0 || rank < 0 || rank >= world)
		return acm::fail(ACM_ERR_ARG, "acm_shard_plan_for: bad arguments");
	const size_t begin = n * (size_t)rank / (size_t)world, end = n * ((size_t)rank + 1) / (size_t)world;
	size_t halo = max_pattern_len > 1 ? (size_t)max_pattern_len - 1 : 0;
	if (halo > begin)
		halo = begin;   // clipped at the start of the text
	out->begin = begin;
	out->end = end;
	out->halo = halo;
	out->load_begin = begin - halo;
	out->load_bytes = end - begin + halo;
	out->offset_shift = (long)(begin - halo);   // local offset + shift = offset in the whole text
	return ACM_OK;
}

extern "C" long acm_merge_planes(const int32_t *all_pat, const int32_t *all_off, int world, size_t plane_capacity,
    int32_t *pat_out, int32_t *off_out, size_t out_capacity, long *last_state)
{
	if (!all_pat || !all_off || world <= 0 || plane_capacity < 2)
		return acm::fail(ACM_ERR_ARG, "acm_merge_planes: bad arguments");
	size_t total = 0;
	long last = 0;
	for (int r = 0; r < world; r++) {   // rank order is position order
		const int32_t *p = all_pat + (size_t)r * plane_capacity, *o = all_off + (size_t)r * plane_capacity;
		const size_t m = (size_t)(p[0] < 0 ? 0 : p[0]);
		if (m + 2 > plane_capacity)
			return acm::fail(ACM_ERR_CAPACITY, "acm_merge_planes: rank %d has %zu records, its planes hold %zu", r, m,
			    plane_capacity - 2);
		if (pat_out && off_out) {
			if (total + m > out_capacity)
				return acm::fail(ACM_ERR_CAPACITY, "acm_merge_planes: %zu records do not fit %zu cells", total + m,
				    out_capacity);
			memcpy(pat_out + total, p + 1, m * sizeof(int32_t));
			memcpy(off_out + total, o + 1, m * sizeof(int32_t));
		}
		total += m;
		last = p[m + 1];   // the last rank's final state is the text's
	}
	if (last_state)
		*last_state = last;
	return (long)total;
}

namespace {

// the four RCCL entry points the gather needs (rccl.h:700-716, ncclGroupStart/End)
struct Rccl {
	int (*group_start)(void) = nullptr;
	int (*group_end)(void) = nullptr;
	int (*send)(const void *, size_t, int, int, void *, void *) = nullptr;
	int (*recv)(void *, size_t, int, int, void *, void *) = nullptr;
	int (*all_gather)(const void *, void *, size_t, int, void *, void *) = nullptr;
	const char *(*error_string)(int) = nullptr;
	bool ok = false;
};

Rccl &rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h)
			h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!h)
			return;
		r.group_start = (int (*)(void))dlsym(h, "ncclGroupStart");
		r.group_end = (int (*)(void))dlsym(h, "ncclGroupEnd");
		r.send = (int (*)(const void *, size_t, int, int, void *, void *))dlsym(h, "ncclSend");
		r.recv = (int (*)(void *, size_t, int, int, void *, void *))dlsym(h, "ncclRecv");
		r.all_gather = (int (*)(const void *, void *, size_t, int, void *, void *))dlsym(h, "ncclAllGather");
		r.error_string = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
		r.ok = r.group_start && r.group_end && r.send && r.recv;
	});
	return r;
}

constexpr int kNcclInt32 = 2;   // ncclInt32 (rccl.h:461)

}  // namespace

extern "C" int acm_gather_planes(void *nccl_comm, int rank, int world, int root, const int32_t *d_pat_plane,
    const int32_t *d_off_plane, size_t plane_capacity, int32_t *d_all_pat, int32_t *d_all_off, void *stream)
{
	if (!nccl_comm || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || !d_pat_plane ||
	    !d_off_plane || plane_capacity < 2 || (rank == root && (!d_all_pat || !d_all_off)))
		return acm::fail(ACM_ERR_ARG, "acm_gather_planes: bad arguments");
	Rccl &r = rccl();
	if (!r.ok)
		return acm::fail(ACM_ERR_NODEV, "acm_gather_planes: librccl.so could not be loaded");
	auto check = [&](int rc, const char *what) -> int {
		if (rc == 0)
			return ACM_OK;
		return acm::fail(ACM_ERR_HIP, "acm_gather_planes: %s: %s", what, r.error_string ? r.error_string(rc) : "RCCL error");
	};
	// one group: every rank sends its two planes to the root, the root posts a receive per rank
	// and plane (its own contribution included: a send to itself is legal inside a group)
	int rc = check(r.group_start(), "ncclGroupStart");
	if (rc != ACM_OK)
		return rc;
	rc = check(r.send(d_pat_plane, plane_capacity, kNcclInt32, root, nccl_comm, stream), "ncclSend");
	if (rc == ACM_OK)
		rc = check(r.send(d_off_plane, plane_capacity, kNcclInt32, root, nccl_comm, stream), "ncclSend");
	if (rank == root)
		for (int peer = 0; peer < world && rc == ACM_OK; peer++) {
			rc = check(r.recv(d_all_pat + (size_t)peer * plane_capacity, plane_capacity, kNcclInt32, peer, nccl_comm, stream),
			    "ncclRecv");
			if (rc == ACM_OK)
				rc = check(r.recv(d_all_off + (size_t)peer * plane_capacity, plane_capacity, kNcclInt32, peer, nccl_comm,
				    stream), "ncclRecv");
		}
	const int end = check(r.group_end(), "ncclGroupEnd");
	return rc != ACM_OK ? rc : end;
}

// The gather as SURVEY 8(e) words it: the record counts first (an all-gather of one cell per rank: the
// header cell of the pattern plane), then sends and receives sized by them -- count + 2 cells of each
// plane (header, records, trailer) instead of plane_capacity.  The counts live on the device, so the
// call waits on `stream` once, for 4 * world bytes, between the two steps; a rank whose planes
// overflowed (count > plane_capacity - 2) sends plane_capacity cells and acm_merge_planes reports it.
// d_counts: int32[world] scratch on every rank.  counts_out (host, may be null): what each rank sent.
extern "C" int acm_gather_planes_sized(void *nccl_comm, int rank, int world, int root, const int32_t *d_pat_plane,
    const int32_t *d_off_plane, size_t plane_capacity, int32_t *d_all_pat, int32_t *d_all_off, int32_t *d_counts,
    int32_t *counts_out, void *stream)
{
	if (!nccl_comm || world <= 0 || world > 4096 || rank < 0 || rank >= world || root < 0 || root >= world || !d_pat_plane ||
	    !d_off_plane || !d_counts || plane_capacity < 2 || (rank == root && (!d_all_pat || !d_all_off)))
		return acm::fail(ACM_ERR_ARG, "acm_gather_planes_sized: bad arguments");
	Rccl &r = rccl();
	if (!r.ok || !r.all_gather)
		return acm::fail(ACM_ERR_NODEV, "acm_gather_planes_sized: librccl.so could not be loaded");
	auto check = [&](int rc, const char *what) -> int {
		if (rc == 0)
			return ACM_OK;
		return acm::fail(ACM_ERR_HIP, "acm_gather_planes_sized: %s: %s", what, r.error_string ? r.error_string(rc) : "RCCL error");
	};
	int rc = check(r.all_gather(d_pat_plane, d_counts, 1, kNcclInt32, nccl_comm, stream), "ncclAllGather");
	if (rc != ACM_OK)
		return rc;
	int32_t counts[4096];
	ACM_HIP_TRY(hipMemcpyAsync(counts, d_counts, (size_t)world * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
	ACM_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	auto cells_of = [&](int32_t m) -> size_t {
		const size_t want = (size_t)(m < 0 ? 0 : m) + 2;
		return want < plane_capacity ? want : plane_capacity;
	};
	if (counts_out)
		memcpy(counts_out, counts, (size_t)world * sizeof(int32_t));
	rc = check(r.group_start(), "ncclGroupStart");
	if (rc != ACM_OK)
		return rc;
	const size_t mine = cells_of(counts[rank]);
	rc = check(r.send(d_pat_plane, mine, kNcclInt32, root, nccl_comm, stream), "ncclSend");
	if (rc == ACM_OK)
		rc = check(r.send(d_off_plane, mine, kNcclInt32, root, nccl_comm, stream), "ncclSend");
	if (rank == root)
		for (int peer = 0; peer < world && rc == ACM_OK; peer++) {
			const size_t n = cells_of(counts[peer]);
			rc = check(r.recv(d_all_pat + (size_t)peer * plane_capacity, n, kNcclInt32, peer, nccl_comm, stream), "ncclRecv");
			if (rc == ACM_OK)
				rc = check(r.recv(d_all_off + (size_t)peer * plane_capacity, n, kNcclInt32, peer, nccl_comm, stream), "ncclRecv");
		}
	const int end = check(r.group_end(), "ncclGroupEnd");
	return rc != ACM_OK ? rc : end;
}
