// The reference's library interface (libacmatch.a: ocl_context.o databuf.o
// ocl_aho_match.o acsmx.o + ocl_prefix_sum.o ocl_compact_array.o, reference
// Makefile:15-20) re-implemented on top of the acm_* layer.
//
// Error behaviour is the reference's: no error returns on the hot path, a
// message on stderr and exit(1) (common.h:12-20).  Return codes exist where
// the reference has them (databuf_add_*, ocl_bitonic_sort).
//
// Semantics that differ on purpose (DESIGN.md "Boundary"):
//   * the scan follows the serial stream semantics (every position exactly
//     once), not the chunk-restart quirks of ahomatch.cl:96-158;
//   * 'mapped' is accepted and ignored: h_* arrays are pinned host memory and
//     every copy really happens;
//   * databuf_process_results returns the match count (the reference falls
//     off the end of the function, databuf.c:787-794).
#include <hip/hip_runtime.h>

#include <malloc.h>
#include <sys/param.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "acm_internal.h"
#include "device_dfa.h"

namespace {

struct Context {
	int device;
};

[[noreturn]] void die(const char *what)
{
	const char *detail = acm_last_error();
	if (detail && *detail)
		fprintf(stderr, "%s: %s\n", what, detail);
	else
		fprintf(stderr, "%s\n", what);
	exit(1);
}

void hip_or_die(hipError_t e, const char *what)
{
	if (e != hipSuccess) {
		fprintf(stderr, "ERROR: %s: %s\n", what, hipGetErrorString(e));
		exit(1);
	}
}

int device_of(const struct clconf *cl)
{
	return cl && cl->ctx ? ((const Context *)cl->ctx)->device : 0;
}

void *pinned(size_t bytes, const char *what)
{
	void *p = nullptr;
	hip_or_die(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault), what);
	return p;
}

cl_mem device_mem(size_t bytes, const char *what)
{
	void *p = nullptr;
	hip_or_die(hipMalloc(&p, bytes ? bytes : 16), what);
	return (cl_mem)p;
}

size_t round16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

// ---------------------------------------------------------------- context ---

extern "C" void clinitctx(struct clconf *cl, int pos, int subpos)
{
	(void)subpos;
	memset(cl, 0, sizeof(*cl));
	int ndev = acm_device_count();
	if (ndev <= 0) {
		fprintf(stderr, "ndevs: no HIP device found\n");
		exit(1);
	}
	if (pos < 0 || pos >= ndev) {
		fprintf(stderr, "invalid dev pos\n");  // ocl_context.c:69-70
		exit(1);
	}
	hip_or_die(hipSetDevice(pos), "ctx");
	hipStream_t s;
	hip_or_die(hipStreamCreate(&s), "queue");
	Context *ctx = new Context{ pos };
	cl->platform = (cl_platform_id)(uintptr_t)1;
	cl->dev = (cl_device_id)(uintptr_t)(pos + 1);
	cl->ctx = (cl_context)ctx;
	cl->queue = (cl_command_queue)s;
	cl->type = 1u << 2;  // CL_DEVICE_TYPE_GPU
}

// nothing to JIT: the code objects are embedded in this library
extern "C" void ocl_aho_match_init(struct clconf *) {}
extern "C" void ocl_aho_match_close(struct clconf *) {}
extern "C" void ocl_prefix_sum_init(struct clconf *) {}
extern "C" void ocl_prefix_sum_close(struct clconf *) {}
extern "C" void ocl_compact_array_init(struct clconf *) {}
extern "C" void ocl_compact_array_close(struct clconf *) {}
extern "C" int ocl_bitonic_sort_init(struct clconf *) { return 0; }
extern "C" int ocl_bitonic_sort_close(struct clconf *) { return 0; }

// ------------------------------------------------------------------- acsm ---

extern "C" acsm_t *acsm_new(void)
{
	acsm_t *a = (acsm_t *)calloc(1, sizeof(acsm_t));
	if (!a) {
		fprintf(stderr, "ERROR! out off memory: acsm_new!\n");  // acsmx.c:87-92
		exit(0);
	}
	a->native = acm_automaton_new();
	if (!a->native) {
		fprintf(stderr, "ERROR! out off memory: acsm_new!\n");
		exit(0);
	}
	return a;
}

extern "C" void acsm_add_pattern(acsm_t *a, unsigned char *pat, int n, int nocase, int offset,
    int depth, void *id, int iid)
{
	(void)nocase;  // stored but ignored by the reference too (acsmx.c:265-275)
	(void)offset;
	(void)depth;
	(void)id;
	if (acm_automaton_add(a->native, pat, n, iid) != ACM_OK)
		die("ERROR: acsm_add_pattern");
	a->num_patterns++;
	if (n > a->max_pattern_len)
		a->max_pattern_len = n;
}

extern "C" void acsm_compile(acsm_t *a)
{
	if (acm_automaton_compile(a->native) != ACM_OK)
		die("ERROR: acsm_compile");
	a->max_states = 1;
	for (int i = 0; i < a->num_patterns; i++) {
		int n = 0;
		acm_automaton_pattern(a->native, i, nullptr, &n, nullptr, nullptr);
		a->max_states += n;
	}
	a->num_states = acm_automaton_num_states(a->native) - 1;  // highest id, as acsmx.c:339-344 leaves it
}

extern "C" void acsm_gen_state_table(acsm_t *a, int mapped, cl_context ctx, cl_command_queue queue)
{
	(void)mapped;
	(void)queue;
	a->num_states += 1;  // acsmx.c:615
	const int device = ctx ? ((Context *)ctx)->device : 0;
	if (acm_dfa_upload(a->native, device, &a->dfa) != ACM_OK)
		die("ERROR: alloc d_trans");
	a->d_trans = (cl_mem)a->dfa->d_cold;
	a->size = acm_dfa_device_bytes(a->dfa);
}

extern "C" acsm_pattern_t *acsm_get_patterns_table(acsm_t *a)
{
	if (!a || !a->native)
		return nullptr;
	const int np = a->num_patterns;
	acsm_pattern_t *t = (acsm_pattern_t *)memalign(0x1000, (np ? np : 1) * sizeof(acsm_pattern_t));
	if (!t)
		return nullptr;
	for (int i = 0; i < np; i++) {
		int iid = 0, n = 0, nxt = -1;
		const unsigned char *bytes = nullptr;
		acm_automaton_pattern(a->native, i, &iid, &n, &bytes, &nxt);
		memset(&t[i], 0, sizeof(t[i]));
		t[i].pattern = (unsigned char *)strndup((const char *)bytes, (size_t)n);
		t[i].casepattern = (unsigned char *)strndup((const char *)bytes, (size_t)n);
		t[i].n = n;
		t[i].iid = iid;
		t[i].index = (unsigned)i;
		t[i].next = nxt >= 0 ? &t[nxt] : nullptr;
	}
	return t;
}

extern "C" int acsm_get_max_pattern_size(acsm_t *a) { return a->max_pattern_len; }
extern "C" int acsm_get_states(acsm_t *a) { return a->num_states; }
extern "C" size_t acsm_get_size(acsm_t *a) { return a->size; }

// drops the host automaton (trie, lists, dense rows); the device DFA stays
extern "C" void acsm_cleanup(acsm_t *a)
{
	acm_automaton_free(a->native);
	a->native = nullptr;
}

extern "C" void acsm_free(acsm_t *a)
{
	if (!a)
		return;
	acm_automaton_free(a->native);
	acm_dfa_release(a->dfa);
	free(a);
}

// ---------------------------------------------------------------- databuf ---

extern "C" struct databuf *databuf_new(size_t max_chunks, size_t max_chunk_size, int max_results,
    int mapped, struct clconf *cl)
{
	struct databuf *db = (struct databuf *)calloc(1, sizeof(struct databuf));
	if (!db) {
		fprintf(stderr, "ERROR: malloc db: %s\n", strerror(errno));
		exit(1);
	}
	hip_or_die(hipSetDevice(device_of(cl)), "databuf_new");
	db->cl = cl;
	db->mapped = mapped;
	db->max_results = max_results;
	db->max_chunks = max_chunks;
	db->max_chunk_size = max_chunk_size;
	db->size = max_chunks * max_chunk_size;
	db->results_comp_size = db->size + 2;   // count + records + last state (databuf.c:99-100)
	db->results2_comp_size = db->size + 2;

	const size_t plane = ((size_t)max_results * max_chunks + 1) * sizeof(int);
	const size_t comp = db->results_comp_size * sizeof(int);

	db->d_data = device_mem(round16(db->size) + 16, "alloc d_data");
	db->d_indices = device_mem((max_chunks + 1) * sizeof(int), "alloc d_indices");
	db->d_sizes = device_mem((max_chunks + 1) * sizeof(int), "alloc d_sizes");
	db->d_results = device_mem(plane, "alloc d_results");
	db->d_results2 = device_mem(plane, "alloc d_results2");
	db->d_prefixsum = device_mem(max_chunks * sizeof(int), "alloc d_prefixsum");
	db->d_results_comp = device_mem(comp, "alloc d_results_comp");
	db->d_results2_comp = device_mem(comp, "alloc d_results2_comp");

	db->h_data = (unsigned char *)pinned(round16(db->size) + 16, "pin h_data");
	db->h_indices = (int *)pinned((max_chunks + 1) * sizeof(int), "pin h_indices");
	db->h_sizes = (int *)pinned((max_chunks + 1) * sizeof(int), "pin h_sizes");
	db->h_results = (int *)pinned(plane, "pin h_results");
	db->h_results2 = (int *)pinned(plane, "pin h_results2");
	db->h_prefixsum = (int *)pinned(max_chunks * sizeof(int), "pin h_prefixsum");
	db->h_results_comp = (int *)pinned(comp, "pin h_results_comp");
	db->h_results2_comp = (int *)pinned(comp, "pin h_results2_comp");

	// scratch: scan pipeline, a packed copy of the text for padded chunk
	// lists, and the packed start offsets
	size_t need = acm_scan_workspace_bytes(nullptr, db->size);
	const size_t scan_need = acm_exclusive_scan_workspace_bytes(max_chunks);
	if (scan_need > need)
		need = scan_need;
	db->ws_bytes = need;
	db->ws = (void *)device_mem(need, "alloc scan workspace");

	db->file_ids = (int *)memalign(0x1000, (max_chunks ? max_chunks : 1) * sizeof(int));
	if (!db->file_ids) {
		fprintf(stderr, "ERROR: malloc file_ids: %s\n", strerror(errno));
		exit(1);
	}
	for (size_t i = 0; i < max_chunks; i++) {  // binary-mode defaults, databuf.c:313-317
		db->h_sizes[i] = (int)max_chunk_size;
		db->h_indices[i] = (int)(max_chunk_size * i);
		db->file_ids[i] = -1;
	}
	return db;
}

// read(2) straight into the next free chunk slot (databuf.c:326-407)
extern "C" int databuf_add_fd(struct databuf *db, int fd, int id, size_t *rd_bytes)
{
	const size_t room = (db->max_chunks - db->chunks) * db->max_chunk_size;
	const ssize_t got = read(fd, &db->h_data[db->h_indices[db->chunks]], room);
	const size_t size = got < 0 ? 0 : (size_t)got;
	*rd_bytes = size;
	if (db->bytes + size > db->size) {
		printf("ERROR: more data in buffer than maximum!\n");
		exit(EXIT_FAILURE);
	}
	if (size == 0)
		return 0;

	const size_t whole = size / db->max_chunk_size;
	for (size_t i = db->chunks; i < db->chunks + whole; i++) {
		db->h_sizes[i] = (int)db->max_chunk_size;
		db->file_ids[i] = id;
	}
	db->chunks += whole;
	const size_t tail = size % db->max_chunk_size;
	if (tail) {  // short last chunk: record its size, zero the rest of the slot
		db->h_sizes[db->chunks] = (int)tail;
		memset(&db->h_data[db->h_indices[db->chunks] + tail], 0, db->max_chunk_size - tail);
		db->file_ids[db->chunks] = id;
		db->chunks++;
	}
	db->bytes = db->chunks * db->max_chunk_size;

	if (db->chunks == db->max_chunks)
		return -1;
	if (size == db->size) {
		printf("MAX_SIZE\n");
		return -2;
	}
	return (int)size;
}

// one chunk per fgets() line (databuf.c:412-481)
extern "C" int databuf_add_fp(struct databuf *db, FILE *fp, int id, int aligned, size_t *rd_bytes,
    size_t *rd_lines)
{
	*rd_bytes = *rd_lines = 0;
	if (db->chunks >= db->max_chunks)
		return -1;
	if (db->bytes >= db->size)
		return -2;
	char *buf = (char *)&db->h_data[db->bytes];
	size_t toread = MIN(db->size - db->bytes, db->max_chunk_size);
	while (fgets(buf, (int)toread, fp) != NULL) {
		const size_t len = strnlen(buf, toread);
		*rd_bytes += len;
		if (len && buf[len - 1] == '\n')
			*rd_lines += 1;

		db->h_indices[db->chunks] = (int)db->bytes;
		db->h_sizes[db->chunks] = (int)len;
		db->file_ids[db->chunks] = id;
		db->chunks += 1;

		const size_t adv = aligned ? round16(len) : len;
		if (aligned) {
			// zero the alignment gap so stale bytes are never scanned
			size_t gap = adv - len;
			if (db->bytes + len + gap > db->size)
				gap = db->size - db->bytes - len;
			memset(&buf[len], 0, gap);
		}
		db->bytes += adv;
		if (db->bytes > db->size)
			db->bytes = db->size;

		if (db->chunks >= db->max_chunks)
			return -1;
		if (db->bytes >= db->size)
			return -2;
		buf = (char *)&db->h_data[db->bytes];
		toread = MIN(db->size - db->bytes, db->max_chunk_size);
	}
	return (int)(db->size - db->bytes);
}

// databuf.c:487-528
extern "C" int databuf_add_chunk(struct databuf *db, char *chunk, size_t len, int id, char aligned)
{
	if (len > db->max_chunk_size)
		return -3;
	if (db->chunks >= db->max_chunks)
		return -1;
	if (db->bytes + len >= db->size)
		return -2;
	memcpy(&db->h_data[db->bytes], chunk, len);
	db->h_indices[db->chunks] = (int)db->bytes;
	db->h_sizes[db->chunks] = (int)len;
	db->file_ids[db->chunks] = id;
	db->chunks += 1;
	if (aligned) {
		const size_t adv = round16(len);
		size_t gap = adv - len;
		if (db->bytes + len + gap > db->size)
			gap = db->size - db->bytes - len;
		memset(&db->h_data[db->bytes + len], 0, gap);  // the reference leaves stale bytes here
		db->bytes += adv;
		if (db->bytes > db->size)
			db->bytes = db->size;
	} else {
		db->bytes += len;
	}
	return (int)(db->size - db->bytes);
}

extern "C" void databuf_reset(struct databuf *db)
{
	db->chunks = 0;
	db->bytes = 0;
	db->scanned = 0;
}

extern "C" void databuf_clear(struct databuf *db)
{
	const size_t plane = ((size_t)db->max_results * db->max_chunks + 1) * sizeof(int);
	memset(db->h_data, 0, db->size);
	memset(db->h_indices, 0, db->max_chunks * sizeof(int));
	memset(db->h_sizes, 0, db->max_chunks * sizeof(int));
	memset(db->h_results, 0, plane);
	memset(db->h_results2, 0, plane);
	memset(db->h_results_comp, 0, db->results_comp_size * sizeof(int));
	memset(db->h_results2_comp, 0, db->results2_comp_size * sizeof(int));
	memset(db->file_ids, 0, db->max_chunks * sizeof(int));
	databuf_reset(db);
}

// databuf.c:574-597 (three blocking writes there; one sync here)
extern "C" void databuf_copy_host_to_device(struct databuf *db, cl_command_queue queue)
{
	hipStream_t s = (hipStream_t)queue;
	hip_or_die(hipSetDevice(device_of(db->cl)), "write d_data");
	hip_or_die(hipMemcpyAsync(db->d_data, db->h_data, db->bytes, hipMemcpyHostToDevice, s),
	    "write d_data");
	hip_or_die(hipMemcpyAsync(db->d_indices, db->h_indices, db->chunks * sizeof(int),
	    hipMemcpyHostToDevice, s), "write d_indices");
	hip_or_die(hipMemcpyAsync(db->d_sizes, db->h_sizes, db->chunks * sizeof(int),
	    hipMemcpyHostToDevice, s), "write d_sizes");
	hip_or_die(hipStreamSynchronize(s), "write d_data");
	db->scanned = 0;
}

// ---- the scan (ocl_aho_match.c:82-131) ----------------------------------------

extern "C" void ocl_aho_match(struct clconf *cl, struct databuf *db, acsm_t *acsm, size_t local_ws,
    int stream)
{
	(void)local_ws;  // the launch shape is the library's business
	(void)stream;    // accepted and ignored, like ocl_aho_match.c:83-90
	if (!acsm || !acsm->dfa)
		die("ocl_aho_match_kernel: ERROR executing kernel: no automaton on the device");
	hipStream_t s = (hipStream_t)cl->queue;
	hip_or_die(hipSetDevice(device_of(cl)), "ocl_aho_match_kernel");
	int32_t *pat_plane = (int32_t *)db->d_results_comp;
	int32_t *off_plane = (int32_t *)db->d_results2_comp;
	const int chunks = (int)db->chunks;

	// is the chunk list one gap-free stream starting at offset 0?
	size_t stream_len = 0;
	bool packed = true;
	for (int i = 0; i < chunks; i++) {
		if ((size_t)db->h_indices[i] != stream_len)
			packed = false;
		stream_len += (size_t)db->h_sizes[i];
	}

	// the scan's scratch depends on the automaton (the databuf was created without one)
	const size_t need = acm_scan_workspace_bytes(acsm->dfa, db->size);
	if (need > db->ws_bytes) {
		hipFree(db->ws);
		db->ws_bytes = need;
		db->ws = (void *)device_mem(need, "alloc scan workspace");
	}
	int rc;
	if (packed) {
		rc = acm_scan_async(acsm->dfa, db->d_data, stream_len, db->last_state, db->ws, db->ws_bytes,
		    pat_plane, off_plane, db->results_comp_size, s);
	} else {
		// pack into a second text buffer owned by the databuf (freed by databuf_free); the packed
		// starts go through a pinned twin that outlives the copy: no synchronisation here
		if (db->pack_text_cap < round16(db->size) + 16) {
			if (db->pack_text)
				hipFree(db->pack_text);
			db->pack_text_cap = round16(db->size) + 16;
			db->pack_text = (void *)device_mem(db->pack_text_cap, "alloc packed text");
		}
		if (db->pack_starts_cap < (size_t)chunks + 1) {
			if (db->pack_starts)
				hipFree(db->pack_starts);
			if (db->h_pack_starts)
				hipHostFree(db->h_pack_starts);
			db->pack_starts_cap = db->max_chunks + 1;
			db->pack_starts = (int *)device_mem(db->pack_starts_cap * sizeof(int32_t), "alloc packed starts");
			db->h_pack_starts = (int *)pinned(db->pack_starts_cap * sizeof(int32_t), "pin packed starts");
		}
		size_t acc = 0;
		for (int i = 0; i < chunks; i++) {
			db->h_pack_starts[i] = (int32_t)acc;
			acc += (size_t)db->h_sizes[i];
		}
		db->h_pack_starts[chunks] = (int32_t)acc;
		hip_or_die(hipMemcpyAsync(db->pack_starts, db->h_pack_starts, ((size_t)chunks + 1) * sizeof(int32_t),
		    hipMemcpyHostToDevice, s), "write packed starts");
		rc = acm_pack_chunks(db->pack_text, db->d_data, (const int32_t *)db->d_indices,
		    (const int32_t *)db->d_sizes, db->pack_starts, chunks, s);
		if (rc == ACM_OK)
			rc = acm_scan_async(acsm->dfa, db->pack_text, stream_len, db->last_state, db->ws,
			    db->ws_bytes, pat_plane, off_plane, db->results_comp_size, s);
		if (rc == ACM_OK)
			rc = acm_remap_offsets(off_plane, stream_len, (const int32_t *)db->d_indices,
			    db->pack_starts, chunks, s);
	}
	if (rc != ACM_OK)
		die("ocl_aho_match_kernel: ERROR executing kernel");
	if (chunks > 0) {
		rc = acm_bucketize(pat_plane, off_plane, (const int32_t *)db->d_indices,
		    (const int32_t *)db->d_sizes, chunks, db->max_results, (int32_t *)db->d_results,
		    (int32_t *)db->d_results2, db->results_comp_size, s);
		if (rc != ACM_OK)
			die("ocl_aho_match_kernel: ERROR executing kernel");
	}
	hip_or_die(hipStreamSynchronize(s), "ocl_aho_match_kernel: ERROR finishing kernel");
	db->scanned = 1;
}

// databuf.c:603-708: bucket planes, last state, and the compact planes
extern "C" void databuf_copy_device_to_host(struct databuf *db, cl_command_queue queue)
{
	hipStream_t s = (hipStream_t)queue;
	hip_or_die(hipSetDevice(device_of(db->cl)), "read d_results");
	const size_t cells = (size_t)db->max_results * db->chunks + 1;
	hip_or_die(hipMemcpyAsync(db->h_results, db->d_results, cells * sizeof(int), hipMemcpyDeviceToHost,
	    s), "read d_results");
	hip_or_die(hipMemcpyAsync(db->h_results2, db->d_results2, cells * sizeof(int),
	    hipMemcpyDeviceToHost, s), "read d_results2");
	hip_or_die(hipMemcpyAsync(db->h_results_comp, db->d_results_comp, 2 * sizeof(int),
	    hipMemcpyDeviceToHost, s), "read d_results_comp");
	hip_or_die(hipStreamSynchronize(s), "read d_results");
	db->last_state = db->h_results[db->chunks * db->max_results];

	if (db->compact) {
		size_t m = (size_t)db->h_results_comp[0];
		if (m > db->results_comp_size - 2)
			m = db->results_comp_size - 2;
		hip_or_die(hipMemcpyAsync(db->h_results_comp, db->d_results_comp, (m + 2) * sizeof(int),
		    hipMemcpyDeviceToHost, s), "read d_results_comp");
		hip_or_die(hipMemcpyAsync(db->h_results2_comp, db->d_results2_comp, (m + 2) * sizeof(int),
		    hipMemcpyDeviceToHost, s), "read d_results2_comp");
		hip_or_die(hipStreamSynchronize(s), "read d_results_comp");
		db->last_state = db->h_results_comp[m + 1];
	}
}

// databuf.c:713-794
extern "C" int databuf_process_results(struct databuf *db,
    int (*cb)(int file_idx, int patrn_idx, int chunk_idx, int offset, void *uarg), void *uarg)
{
	if (db->compact) {
		const int *res = db->h_results_comp, *res2 = db->h_results2_comp;
		const int matches = res[0];
		for (int i = 0; i < matches && (size_t)i < db->results_comp_size - 2; i++) {
			const int off = res2[i + 1];
			const int c = (int)((size_t)off / db->max_chunk_size);
			if (cb)
				cb(db->file_ids[c], res[i + 1], c, off, uarg);
		}
		return matches;
	}
	const int *res = db->h_results, *res2 = db->h_results2;
	const size_t chunks = db->chunks;
	int matches = 0;
	for (size_t i = 0; i < chunks; i++) {
		matches += res[i];
		for (int j = 0; j < res[i] && j < db->max_results - 1; j++) {
			// offset of the last byte + 1, buffer relative (databuf.c:771)
			if (cb)
				cb(db->file_ids[i], res[(j + 1) * chunks + i], (int)i, res2[(j + 1) * chunks + i] + 1,
				    uarg);
		}
	}
	return matches;
}

extern "C" void databuf_free(struct databuf *db, int mapped, cl_command_queue queue)
{
	(void)mapped;
	(void)queue;
	if (!db)
		return;
	hipSetDevice(device_of(db->cl));
	hipHostFree(db->h_data);
	hipHostFree(db->h_indices);
	hipHostFree(db->h_sizes);
	hipHostFree(db->h_results);
	hipHostFree(db->h_results2);
	hipHostFree(db->h_prefixsum);
	hipHostFree(db->h_results_comp);
	hipHostFree(db->h_results2_comp);
	free(db->file_ids);
	hipFree(db->d_data);
	hipFree(db->d_indices);
	hipFree(db->d_sizes);
	hipFree(db->d_results);
	hipFree(db->d_results2);
	hipFree(db->d_prefixsum);
	hipFree(db->d_results_comp);
	hipFree(db->d_results2_comp);
	hipFree(db->ws);
	hipFree(db->pack_text);
	hipFree(db->pack_starts);
	if (db->h_pack_starts)
		hipHostFree(db->h_pack_starts);
	free(db);
}

// ------------------------------------------------- prefix sum / compaction ---

// exclusive scan of the per-chunk counts d_results[0..n) (ocl_prefix_sum.c:218)
extern "C" void ocl_prefix_sum(struct clconf *cl, struct databuf *db, unsigned int n)
{
	hipStream_t s = (hipStream_t)cl->queue;
	hip_or_die(hipSetDevice(device_of(cl)), "ocl_prefix_sum");
	if (acm_exclusive_scan_i32((const int32_t *)db->d_results, (int32_t *)db->d_prefixsum, n, nullptr,
	    db->ws, db->ws_bytes, s) != ACM_OK)
		die("Error: Failed to scan");
	hip_or_die(hipStreamSynchronize(s), "ocl_prefix_sum");
}

// both planes, each followed by a finish (ocl_compact_array.c:129-172)
extern "C" void ocl_compact_array(struct clconf *cl, struct databuf *db, size_t local_ws)
{
	(void)local_ws;
	hipStream_t s = (hipStream_t)cl->queue;
	hip_or_die(hipSetDevice(device_of(cl)), "kernel_compact_array");
	if (db->chunks == 0)
		return;
	if (acm_compact_buckets((int32_t *)db->d_results_comp, (const int32_t *)db->d_results,
	    (const int32_t *)db->d_prefixsum, (int)db->chunks, db->max_results, s) != ACM_OK ||
	    acm_compact_buckets((int32_t *)db->d_results2_comp, (const int32_t *)db->d_results2,
	    (const int32_t *)db->d_prefixsum, (int)db->chunks, db->max_results, s) != ACM_OK)
		die("kernel_compact_array: executing kernel");
	hip_or_die(hipStreamSynchronize(s), "kernel_compact_array: finishing kernel");
}

// ocl_bitonic_sort.c:140-251: 0 = too short, -1 = unsupported shape,
// otherwise the local work size of the last launch (256 there)
extern "C" int ocl_bitonic_sort(struct clconf *cl, cl_mem key_dst, cl_mem val_dst, cl_mem key_src,
    cl_mem val_src, unsigned int batch, unsigned int len, unsigned int dir)
{
	if (len < 2)
		return 0;
	if (len & (len - 1))
		return -1;
	if (len <= 512 && ((size_t)batch * len) % 512 != 0)
		return -1;
	if (hipSetDevice(device_of(cl)) != hipSuccess)
		return -1;
	if (acm_bitonic_sort_u32((uint32_t *)key_dst, (uint32_t *)val_dst, (const uint32_t *)key_src,
	    (const uint32_t *)val_src, batch, len, dir, (void *)cl->queue) != ACM_OK)
		return -1;
	return 256;
}
