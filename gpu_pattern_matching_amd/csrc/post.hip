// Result post-processing kernels: exclusive scan, bucket compaction, bitonic
// key/value sort, compact -> bucket conversion.  gfx950 only.
//
//   exclusive scan   replaces ocl_prefix_sum.c:164-498 + scan_kernel.cl
//                    (Blelloch up-sweep / down-sweep per block in LDS, block
//                    sums scanned recursively, uniform add) -- on int32, not
//                    on float bit patterns.
//   compact_buckets  replaces ocl_compact_array.c:129-172 + compactarray.cl
//   bitonic sort     replaces ocl_bitonic_sort.c:140-251 + BitonicSort.cl,
//                    same comparator network (tie order included)
//   bucketize        produces the planes ahomatch.cl writes
//                    (results[matches*chunks+id], :63-75) from the ordered
//                    compact planes
//   expand matches   all-patterns reporting: (offset,
//                    final state) records -> one record per pattern of the
//                    state's match list; what walking the next chains of
//                    acsm_get_patterns_table (acsmx.c:707-721) is meant to give
#include <hip/hip_runtime.h>

#include "acm_internal.h"
#include "device_dfa.h"

namespace {

// ------------------------------------------------------------------ scan ---

constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanTile = kScanThreads * kScanPerThread;  // 1024 elements per block

// LDS index with one pad word per 32 to keep the tree strides off one bank
__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }

__global__ __launch_bounds__(kScanThreads) void k_scan_block(const int32_t *in, int32_t *out,
    uint32_t n, int32_t *block_sums, int32_t *total_out)
{
	__shared__ int32_t tree[kScanThreads + (kScanThreads >> 5) + 1];
	const int tid = threadIdx.x;
	const uint32_t base = blockIdx.x * kScanTile + tid * kScanPerThread;
	int32_t v[kScanPerThread];
#pragma unroll
	for (int k = 0; k < kScanPerThread; k++)
		v[k] = (base + k < n) ? in[base + k] : 0;
	int32_t run = 0, pre[kScanPerThread];
#pragma unroll
	for (int k = 0; k < kScanPerThread; k++) {
		pre[k] = run;
		run += v[k];
	}
	tree[pad(tid)] = run;

	// up-sweep (reduce)
	int offset = 1;
	for (int d = kScanThreads >> 1; d > 0; d >>= 1) {
		__syncthreads();
		if (tid < d) {
			int ai = offset * (2 * tid + 1) - 1;
			int bi = offset * (2 * tid + 2) - 1;
			tree[pad(bi)] += tree[pad(ai)];
		}
		offset <<= 1;
	}
	__syncthreads();
	if (tid == 0) {
		const int32_t sum = tree[pad(kScanThreads - 1)];
		if (block_sums)
			block_sums[blockIdx.x] = sum;
		if (total_out && gridDim.x == 1)
			*total_out = sum;
		tree[pad(kScanThreads - 1)] = 0;
	}
	// down-sweep
	for (int d = 1; d < kScanThreads; d <<= 1) {
		offset >>= 1;
		__syncthreads();
		if (tid < d) {
			int ai = offset * (2 * tid + 1) - 1;
			int bi = offset * (2 * tid + 2) - 1;
			int32_t t = tree[pad(ai)];
			tree[pad(ai)] = tree[pad(bi)];
			tree[pad(bi)] += t;
		}
	}
	__syncthreads();
	const int32_t excl = tree[pad(tid)];
#pragma unroll
	for (int k = 0; k < kScanPerThread; k++)
		if (base + k < n)
			out[base + k] = excl + pre[k];
}

__global__ __launch_bounds__(kScanThreads) void k_scan_add(int32_t *out, uint32_t n,
    const int32_t *block_offsets)
{
	const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
	const int32_t add = block_offsets[blockIdx.x];
#pragma unroll
	for (int k = 0; k < kScanPerThread; k++)
		if (base + k < n)
			out[base + k] += add;
}

size_t scan_level_elems(size_t n) { return (n + kScanTile - 1) / kScanTile; }

int scan_recursive(const int32_t *in, int32_t *out, size_t n, int32_t *total, int32_t *ws,
    hipStream_t s)
{
	const size_t blocks = scan_level_elems(n);
	if (blocks <= 1) {
		hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(kScanThreads), 0, s, in, out, (uint32_t)n,
		    (int32_t *)nullptr, total);
		ACM_HIP_TRY(hipGetLastError());
		return ACM_OK;
	}
	int32_t *sums = ws;
	hipLaunchKernelGGL(k_scan_block, dim3((uint32_t)blocks), dim3(kScanThreads), 0, s, in, out,
	    (uint32_t)n, sums, (int32_t *)nullptr);
	ACM_HIP_TRY(hipGetLastError());
	int rc = scan_recursive(sums, sums, blocks, total, ws + ((blocks + 63) & ~(size_t)63), s);
	if (rc != ACM_OK)
		return rc;
	hipLaunchKernelGGL(k_scan_add, dim3((uint32_t)blocks), dim3(kScanThreads), 0, s, out, (uint32_t)n,
	    (const int32_t *)sums);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

// --------------------------------------------------------- compact buckets ---

__global__ void k_compact_buckets(int32_t *dst, const int32_t *src, const int32_t *prefix, int len,
    int max_results)
{
	const int gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid == 0) {
		const int32_t total = prefix[len - 1] + src[len - 1];
		dst[0] = total;
		dst[total + 1] = src[(size_t)max_results * len];
	}
	if (gid >= len)
		return;
	const int32_t off = prefix[gid];
	const int32_t m = src[gid];
	for (int i = 0; i < m && i < max_results - 1; ++i)
		dst[off + 1 + i] = src[(size_t)len * (i + 1) + gid];
}

// ------------------------------------------------------------------ sort ---

constexpr unsigned kSortThreads = 1024;
constexpr unsigned kSortTile = 2 * kSortThreads;  // elements per block, in LDS
constexpr unsigned kRefLocalLimit = 512;          // LOCAL_SIZE_LIMIT, ocl_bitonic_sort.c:18

// direction of comparator i (index inside its array) in the merge of 'size'
__device__ __forceinline__ unsigned net_dir(unsigned size, unsigned len, unsigned i, unsigned dir)
{
	if (size == len)
		return dir;
	const unsigned alt = (i & (size >> 1)) != 0;
	return size <= kRefLocalLimit ? alt : (dir ^ alt);
}

__device__ __forceinline__ void cmp_swap(uint32_t &ka, uint32_t &va, uint32_t &kb, uint32_t &vb,
    unsigned d)
{
	if ((ka > kb) == d) {  // BitonicSort.cl:26, :40 -- equal keys swap when d == 0
		uint32_t t = ka; ka = kb; kb = t;
		t = va; va = vb; vb = t;
	}
}

// all stages with size in [size_lo, size_hi] and stride <= kSortThreads, on
// one tile of kSortTile elements held in LDS
__global__ __launch_bounds__(kSortThreads) void k_bitonic_tile(uint32_t *kd, uint32_t *vd,
    const uint32_t *ks, const uint32_t *vs, size_t total, unsigned len, unsigned dir,
    unsigned size_lo, unsigned size_hi)
{
	__shared__ uint32_t lk[kSortTile];
	__shared__ uint32_t lv[kSortTile];
	const unsigned tid = threadIdx.x;
	const size_t tile0 = (size_t)blockIdx.x * kSortTile;
	for (unsigned k = tid; k < kSortTile; k += kSortThreads) {
		const bool in = tile0 + k < total;
		lk[k] = in ? ks[tile0 + k] : 0xFFFFFFFFu;
		lv[k] = in ? vs[tile0 + k] : 0u;
	}
	const size_t comparator = (size_t)blockIdx.x * kSortThreads + tid;  // global comparator id
	const unsigned i = (unsigned)(comparator & (len / 2 - 1));
	for (unsigned size = size_lo; size <= size_hi; size <<= 1) {
		const unsigned d = net_dir(size, len, i, dir);
		unsigned stride = size >> 1;
		if (stride > kSortThreads)
			stride = kSortThreads;
		for (; stride > 0; stride >>= 1) {
			__syncthreads();
			const unsigned pos = 2 * tid - (tid & (stride - 1));
			if (tile0 + pos + stride < total)
				cmp_swap(lk[pos], lv[pos], lk[pos + stride], lv[pos + stride], d);
		}
	}
	__syncthreads();
	for (unsigned k = tid; k < kSortTile; k += kSortThreads)
		if (tile0 + k < total) {
			kd[tile0 + k] = lk[k];
			vd[tile0 + k] = lv[k];
		}
}

// one compare-exchange per thread for a stride that spans tiles
__global__ void k_bitonic_global(uint32_t *k, uint32_t *v, size_t comparators, unsigned len,
    unsigned size, unsigned stride, unsigned dir)
{
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= comparators)
		return;
	const unsigned i = (unsigned)(c & (len / 2 - 1));
	const unsigned d = net_dir(size, len, i, dir);
	const size_t pos = 2 * c - (c & (stride - 1));
	uint32_t ka = k[pos], va = v[pos], kb = k[pos + stride], vb = v[pos + stride];
	cmp_swap(ka, va, kb, vb, d);
	k[pos] = ka; v[pos] = va;
	k[pos + stride] = kb; v[pos + stride] = vb;
}

// -------------------------------------------------------------- bucketize ---

__device__ __forceinline__ uint32_t lower_bound_i32(const int32_t *a, uint32_t n, int32_t key)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (a[mid] < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}

// 'cap' = cells of a plane: a plane whose count exceeds cap - 2 holds the first cap - 2 records
// and its trailer in the last cell (that is how the scan and expand kernels write it)
__global__ void k_bucketize(const int32_t *pat_plane, const int32_t *off_plane,
    const int32_t *indices, const int32_t *sizes, int chunks, int max_results, int32_t *results,
    int32_t *results2, uint32_t cap)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t full = (uint32_t)pat_plane[0];
	const uint32_t m = min(full, cap - 2), tail = min(full + 1, cap - 1);
	if (i == 0) {
		results[(size_t)chunks * max_results] = pat_plane[tail];   // last state
		results2[(size_t)chunks * max_results] = pat_plane[tail];
	}
	if (i >= chunks)
		return;
	const int32_t lo = indices[i], hi = lo + sizes[i];
	const uint32_t r0 = lower_bound_i32(off_plane + 1, m, lo);
	const uint32_t r1 = lower_bound_i32(off_plane + 1, m, hi);
	const int cnt = (int)(r1 - r0);
	results[i] = cnt;
	results2[i] = cnt;
	for (int k = 0; k < cnt && k < max_results - 1; k++) {
		results[(size_t)(k + 1) * chunks + i] = pat_plane[1 + r0 + k];
		results2[(size_t)(k + 1) * chunks + i] = off_plane[1 + r0 + k];
	}
}

// ---------------------------------------------------- chunk list <-> stream ---

// The reference scans chunk by chunk (indices[]/sizes[], databuf.c:326-481);
// chunks need not be adjacent (zero padding after a file's tail chunk or
// after every line in text mode).  The stream we scan is the chunks' bytes
// back to back, so padded buffers are packed first and offsets mapped back.
__global__ __launch_bounds__(256) void k_pack_chunks(uint8_t *dst, const uint8_t *src,
    const int32_t *indices, const int32_t *sizes, const int32_t *packed_start, int chunks)
{
	const int c = blockIdx.x;
	if (c >= chunks)
		return;
	const uint8_t *from = src + indices[c];
	uint8_t *to = dst + packed_start[c];
	const int n = sizes[c];
	for (int k = threadIdx.x; k < n; k += blockDim.x)
		to[k] = from[k];
}

__global__ void k_remap_offsets(int32_t *off_plane, const int32_t *indices,
    const int32_t *packed_start, int chunks, uint32_t max_records)
{
	const uint32_t m = min((uint32_t)off_plane[0], max_records);
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= m)
		return;
	const int32_t off = off_plane[1 + r];
	// last chunk whose packed start is <= off
	uint32_t lo = 0, hi = (uint32_t)chunks;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (packed_start[mid] <= off)
			lo = mid;
		else
			hi = mid;
	}
	off_plane[1 + r] = indices[lo] + (off - packed_start[lo]);
}

}  // namespace

// ------------------------------------------------------------------- ABI ---

extern "C" int acm_pack_chunks(void *d_dst, const void *d_src, const int32_t *d_indices,
    const int32_t *d_sizes, const int32_t *d_packed_start, int chunks, void *stream)
{
	if (!d_dst || !d_src || !d_indices || !d_sizes || !d_packed_start || chunks <= 0)
		return acm::fail(ACM_ERR_ARG, "acm_pack_chunks: bad arguments");
	hipLaunchKernelGGL(k_pack_chunks, dim3(chunks), dim3(256), 0, (hipStream_t)stream, (uint8_t *)d_dst,
	    (const uint8_t *)d_src, d_indices, d_sizes, d_packed_start, chunks);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

extern "C" int acm_remap_offsets(int32_t *d_off_plane, size_t max_records, const int32_t *d_indices,
    const int32_t *d_packed_start, int chunks, void *stream)
{
	if (!d_off_plane || !d_indices || !d_packed_start || chunks <= 0)
		return acm::fail(ACM_ERR_ARG, "acm_remap_offsets: bad arguments");
	if (max_records == 0)
		return ACM_OK;
	hipLaunchKernelGGL(k_remap_offsets, dim3((unsigned)((max_records + 255) / 256)), dim3(256), 0,
	    (hipStream_t)stream, d_off_plane, d_indices, d_packed_start, chunks,
	    (uint32_t)(max_records > 0xFFFFFFFEul ? 0xFFFFFFFEul : max_records));
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

// ---------------------------------------------------------------- expand ---

namespace {

// records of a compact plane pair: [0] = count, [1 .. count], [count + 1] = final state
__global__ void k_expand_count(const int32_t *state_plane, uint32_t max_records, const uint32_t *list_len,
    int32_t *counts)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= max_records)
		return;
	const uint32_t m = min((uint32_t)state_plane[0], max_records);
	counts[i] = i < m ? (int32_t)list_len[state_plane[1 + i]] : 0;
}

__global__ void k_expand_scatter(const int32_t *state_plane, const int32_t *off_plane, uint32_t max_records,
    const uint32_t *list_begin, const uint32_t *list_len, const int32_t *list_pool, const int32_t *first_cell,
    const int32_t *total, int32_t *pat_out, int32_t *off_out, uint32_t out_capacity)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t m_all = (uint32_t)state_plane[0], m = min(m_all, max_records);
	if (i < m) {
		const uint32_t st = (uint32_t)state_plane[1 + i], len = list_len[st], from = list_begin[st];
		const int32_t off = off_plane[1 + i];
		uint32_t d = (uint32_t)first_cell[i];
		for (uint32_t j = 0; j < len; j++, d++)
			if (d + 2 < out_capacity) {
				pat_out[1 + d] = list_pool[from + j];
				off_out[1 + d] = off;
			}
	}
	if (i == 0) {   // header and trailer cells, as the scan writes them
		const uint32_t t = (uint32_t)*total;
		const int32_t last = state_plane[1 + m];   // trailer of the input (a scan with more records than cells clamps it there)
		uint32_t tail = t + 1;
		if (tail > out_capacity - 1)
			tail = out_capacity - 1;
		pat_out[0] = (int32_t)t;
		off_out[0] = (int32_t)t;
		pat_out[tail] = last;
		off_out[tail] = last;
	}
}

size_t expand_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t acm_expand_workspace_bytes(size_t max_records)
{
	return 2 * expand_align((max_records + 1) * sizeof(int32_t)) + 256 +
	       expand_align(acm_exclusive_scan_workspace_bytes(max_records));
}

extern "C" int acm_expand_matches_async(const acm_dfa *d, const int32_t *d_state_plane, const int32_t *d_off_plane,
    size_t max_records, int32_t *d_pat_out, int32_t *d_off_out, size_t out_capacity, void *d_workspace,
    size_t workspace_bytes, void *stream)
{
	if (!d || !d_state_plane || !d_off_plane || !d_pat_out || !d_off_out || out_capacity < 2 ||
	    max_records == 0 || max_records > 0x7FFFFFFFul)
		return acm::fail(ACM_ERR_ARG, "acm_expand_matches_async: bad arguments");
	if (!d_workspace || workspace_bytes < acm_expand_workspace_bytes(max_records))
		return acm::fail(ACM_ERR_ARG, "acm_expand_matches_async: workspace %zu B < required %zu B",
		    workspace_bytes, acm_expand_workspace_bytes(max_records));
	hipStream_t s = (hipStream_t)stream;
	ACM_HIP_TRY(hipSetDevice(d->device));
	char *ws = (char *)d_workspace;
	const size_t plane = expand_align((max_records + 1) * sizeof(int32_t));
	int32_t *counts = (int32_t *)ws, *cells = (int32_t *)(ws + plane), *total = (int32_t *)(ws + 2 * plane);
	void *scan_ws = ws + 2 * plane + 256;
	const uint32_t n = (uint32_t)max_records, blocks = (n + 255) / 256;
	const uint32_t cap = (uint32_t)(out_capacity > 0xFFFFFFFFul ? 0xFFFFFFFFul : out_capacity);
	hipLaunchKernelGGL(k_expand_count, dim3(blocks), dim3(256), 0, s, d_state_plane, n, d->d_list_len, counts);
	ACM_HIP_TRY(hipGetLastError());
	const int rc = acm_exclusive_scan_i32(counts, cells, n, total, scan_ws,
	    acm_exclusive_scan_workspace_bytes(max_records), s);
	if (rc != ACM_OK)
		return rc;
	hipLaunchKernelGGL(k_expand_scatter, dim3(blocks), dim3(256), 0, s, d_state_plane, d_off_plane, n,
	    d->d_list_begin, d->d_list_len, d->d_list_pool, cells, total, d_pat_out, d_off_out, cap);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

extern "C" size_t acm_exclusive_scan_workspace_bytes(size_t n)
{
	size_t cells = 64;
	for (size_t lvl = scan_level_elems(n); lvl > 1; lvl = scan_level_elems(lvl))
		cells += (lvl + 63) & ~(size_t)63;
	cells += 64;
	return cells * sizeof(int32_t);
}

extern "C" int acm_exclusive_scan_i32(const int32_t *d_in, int32_t *d_out, size_t n,
    int32_t *d_total, void *d_workspace, size_t workspace_bytes, void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	if (n == 0) {
		if (d_total)
			ACM_HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(int32_t), s));
		return ACM_OK;
	}
	if (!d_in || !d_out || n > 0xFFFFFFFFul)
		return acm::fail(ACM_ERR_ARG, "acm_exclusive_scan_i32: bad arguments");
	if (scan_level_elems(n) > 1 &&
	    (!d_workspace || workspace_bytes < acm_exclusive_scan_workspace_bytes(n)))
		return acm::fail(ACM_ERR_ARG, "acm_exclusive_scan_i32: workspace too small");
	return scan_recursive(d_in, d_out, n, d_total, (int32_t *)d_workspace, s);
}

extern "C" int acm_compact_buckets(int32_t *d_dst, const int32_t *d_src, const int32_t *d_prefix,
    int len, int max_results, void *stream)
{
	if (!d_dst || !d_src || !d_prefix || len <= 0 || max_results <= 0)
		return acm::fail(ACM_ERR_ARG, "acm_compact_buckets: bad arguments");
	hipLaunchKernelGGL(k_compact_buckets, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	    d_dst, d_src, d_prefix, len, max_results);
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

extern "C" int acm_bitonic_sort_u32(uint32_t *d_key_dst, uint32_t *d_val_dst,
    const uint32_t *d_key_src, const uint32_t *d_val_src, unsigned batch, unsigned len, unsigned dir,
    void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	if (!d_key_dst || !d_val_dst || !d_key_src || !d_val_src)
		return acm::fail(ACM_ERR_ARG, "acm_bitonic_sort_u32: null buffer");
	const size_t total = (size_t)batch * len;
	if (len < 2) {  // "too short to sort": the reference returns before copying anything
		return ACM_OK;
	}
	if (len & (len - 1))
		return -1;  // only power-of-two lengths (ocl_bitonic_sort.c:154-158)
	if (batch == 0)
		return ACM_OK;
	dir = (dir != 0);
	const unsigned tiles = (unsigned)((total + kSortTile - 1) / kSortTile);
	const unsigned first_hi = len < kSortTile ? len : kSortTile;
	hipLaunchKernelGGL(k_bitonic_tile, dim3(tiles), dim3(kSortThreads), 0, s, d_key_dst, d_val_dst,
	    d_key_src, d_val_src, total, len, dir, 2u, first_hi);
	for (unsigned size = 2 * kSortTile; size <= len && size != 0; size <<= 1) {
		for (unsigned stride = size / 2; stride > kSortThreads; stride >>= 1)
			hipLaunchKernelGGL(k_bitonic_global, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256),
			    0, s, d_key_dst, d_val_dst, total / 2, len, size, stride, dir);
		hipLaunchKernelGGL(k_bitonic_tile, dim3(tiles), dim3(kSortThreads), 0, s, d_key_dst, d_val_dst,
		    (const uint32_t *)d_key_dst, (const uint32_t *)d_val_dst, total, len, dir, size, size);
	}
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}

extern "C" int acm_bucketize(const int32_t *d_pat_plane, const int32_t *d_off_plane,
    const int32_t *d_indices, const int32_t *d_sizes, int chunks, int max_results,
    int32_t *d_results, int32_t *d_results2, size_t plane_capacity, void *stream)
{
	if (!d_pat_plane || !d_off_plane || !d_indices || !d_sizes || !d_results || !d_results2 ||
	    chunks <= 0 || max_results <= 0 || plane_capacity < 2)
		return acm::fail(ACM_ERR_ARG, "acm_bucketize: bad arguments");
	hipLaunchKernelGGL(k_bucketize, dim3((chunks + 255) / 256), dim3(256), 0, (hipStream_t)stream,
	    d_pat_plane, d_off_plane, d_indices, d_sizes, chunks, max_results, d_results, d_results2,
	    (uint32_t)(plane_capacity > 0xFFFFFFFFul ? 0xFFFFFFFFul : plane_capacity));
	ACM_HIP_TRY(hipGetLastError());
	return ACM_OK;
}
