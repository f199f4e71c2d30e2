/*
 * acmatch.h -- C ABI of libacmatch.so, the MI355X-native (gfx950, HIP)
 * Aho-Corasick matcher.
 *
 * Two layers live in this one header:
 *
 *  (1) acm_*  : the native boundary.  Plain pointers and sizes, explicit
 *               device pointers + stream, integer status codes.  This is
 *               what an FFI (ctypes, cgo, JNI) binds, and what bench.py uses.
 *
 *  (2) the reference's own names (acsm_*, databuf_*, ocl_aho_match*,
 *      ocl_prefix_sum*, ocl_compact_array*, ocl_bitonic_sort*, clinitctx):
 *               same names, arity, argument meaning and error behaviour as
 *               the OpenCL library they replace, so ocl_worker.c /
 *               ocl_aho_grep.c recompile against include/compat/ unchanged.
 *               Each declaration cites the reference interface it replaces.
 *
 * Handle mapping for layer (2):  cl_mem = HIP device pointer,
 * cl_command_queue = hipStream_t, cl_context = opaque per-device context,
 * cl_device_id / cl_platform_id = HIP device ordinal + 1 cast to a pointer.
 */
#ifndef ACMATCH_H_
#define ACMATCH_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ====================================================================== */
/* (1) native boundary                                                    */
/* ====================================================================== */

enum {
	ACM_OK = 0,
	ACM_ERR_ARG = -1,	/* bad argument / misuse                       */
	ACM_ERR_NOMEM = -2,	/* host or device allocation failed            */
	ACM_ERR_HIP = -3,	/* a HIP runtime call failed                   */
	ACM_ERR_NODEV = -4,	/* no usable gfx950 device                     */
	ACM_ERR_LIMIT = -5,	/* automaton / buffer exceeds a design limit   */
	ACM_ERR_IO = -6,	/* pattern file could not be opened            */
	ACM_ERR_PARSE = -7,	/* malformed pattern line                      */
	ACM_ERR_CAPACITY = -8	/* result planes too small for the matches     */
};

/* text of the last error raised on the calling thread ("" if none) */
const char *acm_last_error(void);
const char *acm_strerror(int code);
/* library version / build string, e.g. "acmatch 0.1 gfx950" */
const char *acm_version(void);
/* number of visible HIP devices (0 when there is none, never an error) */
int acm_device_count(void);

/* ---------------------------------------------------------------------- */
/* host-side automaton: pattern list -> acsmx-compatible DFA               */
/* replaces acsm_new/add_pattern/compile/gen_state_table (acsmx.h:96-141)  */
/* ---------------------------------------------------------------------- */
typedef struct acm_automaton acm_automaton;

acm_automaton *acm_automaton_new(void);
void acm_automaton_free(acm_automaton *);

/* append one pattern; its index is the number of patterns added before it
 * (acsmx.c:535).  n may be 0 (an empty pattern never reports). */
int acm_automaton_add(acm_automaton *, const unsigned char *bytes, int n,
    int iid);

/* pattern-file parser of ocl_worker.c:74-145: one pattern per line, optional
 * "ID pattern" categorical form decided on the first line, one pair of
 * surrounding double quotes stripped, hex => printable hex (utils.c:32-54),
 * max_len = -m limit in bytes or -1.  Returns patterns added or an error. */
int acm_automaton_load_file(acm_automaton *, const char *path, int hex,
    int max_len);

/* trie -> fail links -> full DFA, reference state numbering preserved
 * (acsmx.c:552-594) and a BFS renumbering derived for the device. */
int acm_automaton_compile(acm_automaton *);

int acm_automaton_num_patterns(const acm_automaton *);
int acm_automaton_max_pattern_len(const acm_automaton *);
/* byte classes of a compiled automaton: bytes that occur in no pattern share
 * one DFA column, every other byte has its own.  256 = every byte its own
 * (no compression).  class_of, if not NULL, gets the 256-entry byte -> class
 * map.  The chain pipeline's planes have one cell per class. */
int acm_automaton_byte_classes(const acm_automaton *, uint8_t *class_of);
/* The LDS-resident form of a small automaton (full rows for the shallow states, a 4-byte record
 * for every other: csrc/compact_tables.h), built on the host and checked transition by transition
 * against the dense DFA -- no device needed.  lds_bytes: what the image may take (0: a CU's 160 KiB
 * less a margin).  Returns 1 if the set qualifies and every (state, byte) transition agrees, 0 if it
 * does not qualify (too many states or byte classes, or the rows that must be full do not fit),
 * negative on error.  stats, if not NULL, gets 9 words: states, classes, rows, side entries, image
 * bytes, rows beyond the mandatory ones, simple records, side entries ending in a row, side entries
 * deferring to the fail state. */
int acm_compact_selftest(const acm_automaton *, uint32_t lds_bytes, uint32_t *stats);
/* tuning aid, host only: walks text through the LDS form from the root; counts[5] = steps, steps
 * decided by the record or a row alone, by one side entry, by more than one hop, final states entered.
 * Returns 1, or 0 if the set does not qualify. */
int acm_compact_profile(const acm_automaton *, const unsigned char *text, size_t n, uint64_t *counts);
/* states as acsm_get_states reports them after acsm_gen_state_table */
int acm_automaton_num_states(const acm_automaton *);
/* bytes of the reference's serialised table: states * 2 * 256 * 4 */
size_t acm_automaton_reference_table_bytes(const acm_automaton *);
/* write the reference-format table [states][2][256] int32 (acsmx.c:640-658);
 * cells the reference leaves uninitialised are written as 0 */
int acm_automaton_export_reference_table(const acm_automaton *, int32_t *dst);
/* pattern i: iid, length, bytes (borrowed pointer), index of the next
 * pattern chained to it by acsm_get_patterns_table (acsmx.c:707-721) or -1 */
int acm_automaton_pattern(const acm_automaton *, int index, int *iid, int *n,
    const unsigned char **bytes, int *next_chained);
/* head-of-match-list pattern index of reference state s, or -1 */
int acm_automaton_state_output(const acm_automaton *, int ref_state);
/* every pattern that ends where the walk enters ref_state, in the order of the
 * state's match list (acsmx.c:299-312, :417-429; the first one is what
 * acm_automaton_state_output returns).  Writes at most cap indices, returns
 * the length of the list (0 for a non-final state, -1 on a bad argument). */
int acm_automaton_state_matches(const acm_automaton *, int ref_state, int32_t *out, int cap);

/* ---------------------------------------------------------------------- */
/* device-resident DFA                                                     */
/* replaces the d_trans upload of acsm_gen_state_table (acsmx.c:618-666)   */
/* ---------------------------------------------------------------------- */
typedef struct acm_dfa acm_dfa;

/* builds the HBM/LDS layout and uploads it to HIP device 'device' */
int acm_dfa_upload(const acm_automaton *, int device, acm_dfa **out);
void acm_dfa_release(acm_dfa *);
size_t acm_dfa_device_bytes(const acm_dfa *);
int acm_dfa_hot_rows(const acm_dfa *);	/* rows staged in LDS by the scan */
int acm_dfa_device(const acm_dfa *);

/* ---------------------------------------------------------------------- */
/* scan pipeline on caller-owned device memory                             */
/* replaces ocl_aho_match + ocl_prefix_sum + ocl_compact_array             */
/* (ocl_aho_match.h:28-29, ocl_prefix_sum.h:21-22, ocl_compact_array.h:21) */
/* ---------------------------------------------------------------------- */

/* bytes of scratch the pipeline needs to scan up to max_text bytes */
size_t acm_scan_workspace_bytes(const acm_dfa *, size_t max_text);

/*
 * Scan d_text[0..n) as one byte stream starting from reference state
 * init_state and produce, in position order, one record per text position
 * whose transition enters a final state:
 *
 *   d_pat_plane[0] = m            d_off_plane[0] = m
 *   d_pat_plane[1..m] = pattern index (head of match list, acsmx.c:650)
 *   d_off_plane[1..m] = offset of the LAST byte of the match
 *   d_*_plane[m+1] = state after the last byte, reference numbering
 *
 * i.e. the compact layout of compactarray.cl:49-55 / databuf.c:656-682.
 * plane_capacity counts int32 cells per plane and must be >= 2; when
 * m + 2 > plane_capacity the first plane_capacity-2 records are stored,
 * cell [0] still holds the full m, the state goes to cell
 * [plane_capacity-1], and ACM_ERR_CAPACITY is reported by acm_scan_finish.
 *
 * d_text must be 16-byte aligned and readable up to n rounded up to 16.
 * Everything is enqueued on 'stream' (a hipStream_t, NULL = default stream);
 * nothing is synchronised.  n <= 2^31 - 17.
 */
int acm_scan_async(const acm_dfa *, const void *d_text, size_t n,
    long init_state, void *d_workspace, size_t workspace_bytes,
    int32_t *d_pat_plane, int32_t *d_off_plane, size_t plane_capacity,
    void *stream);

/*
 * Shard form of acm_scan_async for texts split across devices (or rounds):
 * the first 'halo' bytes of d_text are context only -- they warm the state
 * up, records ending inside them are dropped -- and every reported offset is
 * shifted by offset_shift (e.g. shard_base - halo, to report offsets in the
 * coordinates of the whole text).  With halo >= max_pattern_len - 1 bytes of
 * real preceding text and init_state 0, the records equal those of the
 * serial scan of the whole text restricted to this shard.
 */
int acm_scan_shard_async(const acm_dfa *, const void *d_text, size_t n,
    size_t halo, long offset_shift, long init_state, void *d_workspace,
    size_t workspace_bytes, int32_t *d_pat_plane, int32_t *d_off_plane,
    size_t plane_capacity, void *stream);

/* tuning knobs for acm_scan_async; 0 = automatic.  chain_bytes: bytes each
 * lane walks per tile (power of two, 16..256).  Returns the value in use. */
int acm_scan_set_chain_bytes(acm_dfa *, int chain_bytes);

/*
 * Everything acm_scan_shard_async takes, plus two optional hipEvent_t for
 * keeping several batches in flight on different streams (what the
 * reference does with its -w workers, one queue each): the first kernel
 * of a pipeline (chain: the walk, sparse: the bulk pass) is the stage that
 * fills the device, so consecutive batches can be chained through it --
 * batch k+1's first kernel waits for wait_before_walk (recorded by batch k
 * as record_after_walk) -- while everything behind it runs concurrently
 * with the next batch's first kernel.  Optional: independent streams alone
 * overlap as well, and that is what bench.py measures.
 */
typedef struct acm_scan_batch {
	const void *d_text;
	size_t n;
	size_t halo;
	long offset_shift;
	long init_state;
	void *d_workspace;
	size_t workspace_bytes;
	int32_t *d_pat_plane;
	int32_t *d_off_plane;
	size_t plane_capacity;
	void *stream;
	void *wait_before_walk;		/* hipEvent_t or NULL */
	void *record_after_walk;	/* hipEvent_t or NULL */
	int report;			/* ACM_REPORT_* */
	int profile;			/* non-zero: time this batch's kernels with
					 * events as acm_scan_profile_enable does
					 * for all (acm_scan_profile_read collects) */
	/* The state to start in, handed over on the device: the pattern plane of
	 * the scan this one continues (its trailer cell, found behind its header
	 * cell, holds that scan's final state: databuf.c:622, ahomatch.cl:42-43)
	 * and the capacity that plane was scanned with.  NULL: init_state above.
	 * The scan it names must be in front of this one on the same stream, or
	 * complete.  No host read between consecutive buffers of a worker; such a
	 * batch has its launches to itself (it does not join a launch group). */
	const int32_t *d_init_plane;
	size_t init_plane_capacity;
} acm_scan_batch;

/* What the pattern plane of a scan holds per record.
 *   HEAD   the pattern index the reference reports: the head of the final
 *          state's match list (acsmx.c:650)
 *   STATE  the final state itself (reference numbering), for
 *          acm_expand_matches_async: all-patterns reporting, SURVEY 8(f) row 4 */
enum { ACM_REPORT_HEAD = 0, ACM_REPORT_STATE = 1 };

/* All-patterns reporting.  Input: the planes of a scan enqueued with
 * report = ACM_REPORT_STATE (at most max_records records are looked at).
 * Output, same cell layout ([0] = count, records, trailer = final state): one
 * record per pattern of each final state's match list -- every pattern that
 * ends at that offset, in list order, offsets ascending.  The reference never
 * reports more than the head (ocl_aho_grep.c:276-279 has the table for it,
 * acsmx.c:707-721); off unless asked for.  Stream-ordered, no host sync. */
size_t acm_expand_workspace_bytes(size_t max_records);
int acm_expand_matches_async(const acm_dfa *, const int32_t *d_state_plane,
    const int32_t *d_off_plane, size_t max_records, int32_t *d_pat_out,
    int32_t *d_off_out, size_t out_capacity, void *d_workspace,
    size_t workspace_bytes, void *stream);

int acm_scan_batch_async(const acm_dfa *, const acm_scan_batch *);

/* count batches with one call, enqueued in array order: what a worker pool
 * issues per round, without a trip through the FFI per batch.  Stops at the
 * first batch that fails and returns its status (the batches before it stay
 * enqueued).
 *
 * Grouping: consecutive batches that take the sparse pipeline, have the same
 * stream and size, each its OWN workspace and planes and no events, are
 * enqueued up to acm_scan_set_max_group() at a time as ONE set
 * of three launches -- the kernels' fixed costs (launch, filter fill, the
 * check kernel's chains of dependent loads) are paid per group instead of per
 * batch.  Results are those of enqueueing the batches one by one; they become
 * visible in stream order when the group's last kernel has run.  Batches that
 * share a workspace are never grouped. */
int acm_scan_batches_async(const acm_dfa *, const acm_scan_batch *batches, size_t count);
/* batches per group, 1 (never group) .. 16 (the default).  Returns the value in use;
 * 0 or negative only queries. */
int acm_scan_set_max_group(acm_dfa *, int batches);

/* independent chains each lane interleaves in the walk kernel: 2 or 4.
 * Returns the value in use. */
int acm_scan_set_chains_per_lane(acm_dfa *, int chains);

/* number of kernels the chain pipeline enqueues for a non-empty text of up to
 * 64 MiB (one more above that) */
int acm_scan_kernel_count(void);

/* A scan whose arguments repeat (same text buffer, size, workspace, planes,
 * init_state ... -- a worker cycling through its staging buffers) is captured
 * into a HIP graph the second time it is seen and replayed from then on: one
 * hipGraphLaunch instead of a row of kernel launches.  Off by default (on
 * MI355X / ROCm 7.2 it saves a few microseconds of host time per scan and
 * nothing on the GPU); never used with the NULL stream, profiling or the
 * event fields of acm_scan_batch.  enable = 0 / 1, -1 only queries.  Returns
 * the setting in use. */
int acm_scan_set_graphs(acm_dfa *, int enable);

/* Which pipeline acm_scan_*_async runs.  Both produce the same planes.
 *   CHAIN   speculative chains (any pattern set)
 *   SPARSE  strided 3-gram sieve + exact checks + trie-path followers; needs
 *           every pattern to have at least 3 bytes (otherwise CHAIN is used).
 *           Exact on any text, three launches, no fallback; slow on a text
 *           that is dense in matches
 *   AUTO    SPARSE when the pattern set allows it -- adaptively: when half of
 *           the last 16 sparse batches held more than a record per 128 bytes
 *           or more than a flagged sample per 48 (the worst of real binaries) the next 64
 *           go to the chain pipeline; then the sparse one is tried again, 4
 *           batches at a time, and every bad look quadruples the chain
 *           pipeline's share (up to 4096 batches)
 * Returns the mode in use after the call; acm_scan_mode(d, -1) only queries. */
enum { ACM_SCAN_MODE_AUTO = 0, ACM_SCAN_MODE_CHAIN = 1, ACM_SCAN_MODE_SPARSE = 2 };
int acm_scan_set_mode(acm_dfa *, int mode);
/* 1 when the pattern set qualifies for the sparse pipeline */
int acm_scan_sparse_eligible(const acm_dfa *);
/* 1 when the chain pipeline walks this set with the whole automaton in LDS (csrc/lds_walk.hip: small
 * alphabet, at most 16384 states, patterns of at most 33 bytes); 0: hot rows in LDS + cold plane in HBM */
int acm_scan_lds_resident(const acm_dfa *);
/* 1 when consecutive batches of one size handed to acm_scan_batches_async can share their kernel
 * launches in the current mode: the sparse pipeline's batches, and the chain pipeline's when the
 * automaton is LDS-resident */
int acm_scan_group_capable(const acm_dfa *);
/* after a scan of n bytes with this workspace has been enqueued on stream:
 * waits for the stream and says which pipeline produced the planes --
 * ACM_SCAN_MODE_CHAIN or ACM_SCAN_MODE_SPARSE (0xDEAD: the sparse kernels found
 * their own workspace inconsistent and wrote empty planes; cannot happen) */
int acm_scan_path_taken(const acm_dfa *, const void *d_workspace, size_t n, void *stream);

/* in-line timing with HIP events on the launch stream: when enabled, every
 * acm_scan_async records an event before its first kernel, after its first
 * kernel (chain: the walk; sparse: the bulk kernel k_sieve) and after its
 * last.
 * acm_scan_profile_read waits for the recorded events, returns the summed
 * milliseconds of the first kernel, the second, and the whole pipeline over
 * 'launches' calls, and resets the accumulation.  (Sparse pipeline: first =
 * the bulk kernel k_sieve, second = check + emit.)  Safe to use while other
 * threads enqueue scans on other streams. */
int acm_scan_profile_enable(acm_dfa *, int enable);
int acm_scan_profile_read(acm_dfa *, double *first_ms, double *second_ms,
    double *pipeline_ms, int *launches);

/* ---------------------------------------------------------------------- */
/* standalone result post-processing ops (device pointers, async)          */
/* ---------------------------------------------------------------------- */

/* exclusive int32 prefix sum (work-efficient Blelloch up/down sweep in LDS,
 * recursive over block sums).  Replaces ocl_prefix_sum.c:164-221 +
 * scan_kernel.cl.  d_in may equal d_out.  d_total (optional) gets the sum. */
size_t acm_exclusive_scan_workspace_bytes(size_t n);
int acm_exclusive_scan_i32(const int32_t *d_in, int32_t *d_out, size_t n,
    int32_t *d_total, void *d_workspace, size_t workspace_bytes, void *stream);

/* bucket planes -> dense array; compactarray.cl:40-68 cell for cell:
 * dst[0]=total, dst[1..]=cells, dst[total+1]=src[max_results*len] */
int acm_compact_buckets(int32_t *d_dst, const int32_t *d_src,
    const int32_t *d_prefix, int len, int max_results, void *stream);

/* key/value bitonic sort on uint32 keys, same network as BitonicSort.cl
 * (tie order included); len must be a power of two; returns 0, or -1 for an
 * unsupported length like ocl_bitonic_sort.c:154-165.  src may equal dst. */
int acm_bitonic_sort_u32(uint32_t *d_key_dst, uint32_t *d_val_dst,
    const uint32_t *d_key_src, const uint32_t *d_val_src, unsigned batch,
    unsigned len, unsigned dir, void *stream);

/* compact planes (position ordered) -> the reference's bucket planes
 * results/results2 [max_results][chunks] + trailer (ahomatch.cl:63-75,
 * :90-93, :160-162 layout; databuf.c:747-782 reads it).  plane_capacity =
 * cells of the compact planes: a plane whose count exceeds plane_capacity - 2
 * holds the first plane_capacity - 2 records and its trailer in the last cell. */
int acm_bucketize(const int32_t *d_pat_plane, const int32_t *d_off_plane,
    const int32_t *d_indices, const int32_t *d_sizes, int chunks,
    int max_results, int32_t *d_results, int32_t *d_results2,
    size_t plane_capacity, void *stream);

/* chunk list -> contiguous stream and back.  The reference scans chunk by
 * chunk (indices[]/sizes[], databuf.c:326-481) and chunks may be separated by
 * zero padding; acm_pack_chunks copies chunk i to d_dst + d_packed_start[i],
 * acm_remap_offsets rewrites the offsets of a compact plane (scanned over the
 * packed stream) into offsets of the original buffer.  max_records bounds
 * the launch and the records touched (at most plane capacity - 2); the
 * record count is read from d_off_plane[0] on the device. */
int acm_pack_chunks(void *d_dst, const void *d_src, const int32_t *d_indices,
    const int32_t *d_sizes, const int32_t *d_packed_start, int chunks,
    void *stream);
int acm_remap_offsets(int32_t *d_off_plane, size_t max_records,
    const int32_t *d_indices, const int32_t *d_packed_start, int chunks,
    void *stream);

/* ---------------------------------------------------------------------- */
/* multi-GPU: shard plan and the gather of the match planes                */
/* (new functionality: the reference takes one -D, ocl_aho_grep.c:498-502) */
/* ---------------------------------------------------------------------- */

/* One process (or thread) per device.  The text is cut by range: rank g of
 * 'world' owns [begin, end) of the n bytes, loads load_bytes from load_begin
 * on (its range and, in front of it, a halo of max_pattern_len - 1 bytes) and
 * scans them with acm_scan_shard_async(halo, offset_shift): the records it
 * reports are the serial scan's records that end in its range, with offsets
 * in the coordinates of the whole text.  The DFA is replicated; nothing is
 * exchanged during the scan. */
typedef struct acm_shard_plan {
	size_t begin, end;	/* the rank's share of the text              */
	size_t halo;		/* context bytes in front of it               */
	size_t load_begin;	/* first byte the rank reads (begin - halo)   */
	size_t load_bytes;	/* bytes it scans, halo included              */
	long offset_shift;	/* local offset + shift = offset in the text  */
} acm_shard_plan;
int acm_shard_plan_for(size_t n, int world, int rank, int max_pattern_len,
    acm_shard_plan *out);

/* The one exchange: every rank's compact planes (plane_capacity cells each)
 * go to 'root' over RCCL -- nccl_comm is the caller's ncclComm_t; the sends and
 * receives form one group on 'stream', point to point, nothing is
 * synchronised.  On the root d_all_pat / d_all_off receive them rank-major,
 * [world][plane_capacity]; other ranks may pass NULL.  librccl.so is loaded on
 * first use. */
int acm_gather_planes(void *nccl_comm, int rank, int world, int root,
    const int32_t *d_pat_plane, const int32_t *d_off_plane,
    size_t plane_capacity, int32_t *d_all_pat, int32_t *d_all_off,
    void *stream);

/* The same with sized messages: an all-gather of the ranks' record
 * counts (the header cells), then count + 2 cells of each plane per rank instead
 * of plane_capacity -- 7 MB instead of 2 x 8 MB for the sentiment planes.  The
 * counts are on the device: the call waits on 'stream' once, for 4 * world
 * bytes, between the two steps.  d_counts: int32[world] device scratch on every
 * rank; counts_out: host int32[world] or NULL.  A rank whose count exceeds
 * plane_capacity - 2 sends plane_capacity cells; acm_merge_planes then returns
 * ACM_ERR_CAPACITY for it (the single-GPU contract reports such an overflow
 * through the count in the header cell: the merge has no room for the records). */
int acm_gather_planes_sized(void *nccl_comm, int rank, int world, int root,
    const int32_t *d_pat_plane, const int32_t *d_off_plane,
    size_t plane_capacity, int32_t *d_all_pat, int32_t *d_all_off,
    int32_t *d_counts, int32_t *counts_out, void *stream);

/* Host side of the root, after the gather has been copied back: rank order
 * is position order, so the ranks' records back to back are the text's
 * ordered list.  Writes them to pat_out / off_out (either may be NULL to only
 * count), returns how many there are or an error; *last_state = the final
 * state of the last rank's shard = the text's. */
long acm_merge_planes(const int32_t *all_pat, const int32_t *all_off,
    int world, size_t plane_capacity, int32_t *pat_out, int32_t *off_out,
    size_t out_capacity, long *last_state);

/* ---------------------------------------------------------------------- */
/* device-runtime helpers for FFI hosts without a HIP binding              */
/* ---------------------------------------------------------------------- */
int acm_rt_set_device(int device);
int acm_rt_malloc(void **out, size_t bytes);
int acm_rt_free(void *p);
int acm_rt_host_alloc(void **out, size_t bytes);	/* pinned */
int acm_rt_host_free(void *p);
int acm_rt_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int acm_rt_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int acm_rt_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int acm_rt_memset(void *dst, int value, size_t bytes, void *stream);
int acm_rt_stream_create(void **out);
int acm_rt_stream_destroy(void *stream);
int acm_rt_stream_sync(void *stream);
int acm_rt_device_sync(void);
int acm_rt_event_create(void **out);
int acm_rt_event_destroy(void *ev);
int acm_rt_event_record(void *ev, void *stream);
int acm_rt_event_sync(void *ev);
int acm_rt_event_elapsed_ms(void *start, void *stop, float *ms);
int acm_rt_device_info(int device, char *name, int name_cap, int *cus,
    size_t *mem_bytes, int *lds_per_cu);

/* ====================================================================== */
/* (2) the reference's interface                                           */
/* ====================================================================== */

/* OpenCL handle names, declared exactly as <CL/cl.h> does so that a caller
 * that also includes the Khronos headers sees compatible typedefs. */
#ifndef __OPENCL_CL_H
typedef struct _cl_platform_id *cl_platform_id;
typedef struct _cl_device_id *cl_device_id;
typedef struct _cl_context *cl_context;
typedef struct _cl_command_queue *cl_command_queue;
typedef struct _cl_mem *cl_mem;
typedef struct _cl_program *cl_program;
typedef struct _cl_kernel *cl_kernel;
typedef uint64_t cl_device_type;
#endif

/* ---- ocl_context.h:8-38 ------------------------------------------------ */
struct clconf {
	cl_platform_id platform;
	cl_device_id dev;
	cl_context ctx;		/* per-device context of this library      */
	cl_command_queue queue;	/* a hipStream_t                           */

	cl_program program_aho_match;	/* unused: code objects are embedded */
	cl_kernel kernel_aho_match;
	cl_program program_prefixsum;
	cl_kernel kernel_prescan;
	cl_kernel kernel_prescan_store_sum;
	cl_kernel kernel_prescan_store_sum_non_power_of_two;
	cl_kernel kernel_prescan_non_power_of_two;
	cl_kernel kernel_uniform_add;
	cl_program program_compact_array;
	cl_kernel kernel_compact_array;

	cl_device_type type;
};

/* pick the pos-th device, create context + in-order queue
 * (ocl_context.c:18-85; subpos is ignored there too) */
void clinitctx(struct clconf *, int pos, int subpos);

/* ---- acsmx.h:44-196 ---------------------------------------------------- */
#define ALPHABET_SIZE 256
#define ACSM_FAIL_STATE -1

struct _acsm_pattern {
	struct _acsm_pattern *next;
	unsigned char *pattern;
	unsigned char *casepattern;
	int n;
	int nocase;
	int offset;
	int depth;
	void *id;
	int iid;
	unsigned int index;
};
typedef struct _acsm_pattern acsm_pattern_t;

struct _acsm_state_table {	/* kept for source compatibility; unused */
	int next_state[ALPHABET_SIZE];
	int fail_state;
	int num_finals;
	acsm_pattern_t *match_list;
};
typedef struct _acsm_state_table acsm_state_table_t;

struct _acsm {
	int max_states;
	int num_states;
	int max_pattern_len;
	size_t size;
	acsm_pattern_t *patterns;	/* always NULL here: see 'native'   */
	int num_patterns;
	acsm_state_table_t *state_table;	/* always NULL              */
	int *h_trans;			/* NULL: no host copy is kept       */
	cl_mem d_trans;			/* device table (cold plane)        */
	acm_automaton *native;		/* host automaton                   */
	acm_dfa *dfa;			/* device DFA after gen_state_table */
};
typedef struct _acsm acsm_t;

acsm_t *acsm_new(void);						/* acsmx.h:101 */
void acsm_add_pattern(acsm_t *, unsigned char *, int n, int nocase,
    int offset, int depth, void *id, int iid);			/* :118 */
void acsm_compile(acsm_t *);					/* :127 */
void acsm_gen_state_table(acsm_t *, int mapped, cl_context,
    cl_command_queue);						/* :139 */
acsm_pattern_t *acsm_get_patterns_table(acsm_t *);		/* :152 */
int acsm_get_max_pattern_size(acsm_t *);			/* :162 */
int acsm_get_states(acsm_t *);					/* :172 */
size_t acsm_get_size(acsm_t *);					/* :182 */
void acsm_cleanup(acsm_t *);					/* :191 */
void acsm_free(acsm_t *);					/* :200 */

/* ---- databuf.h:9-174 --------------------------------------------------- */
#define MAX_RESULTS 16

struct databuf {
	unsigned char *h_data;
	int *h_indices;
	int *h_sizes;
	int *h_results;
	int *h_results2;
	int *h_prefixsum;
	int *h_results_comp;
	int *h_results2_comp;

	size_t results_comp_size;
	size_t results2_comp_size;

	int *file_ids;
	int mapped;
	int max_results;
	long last_state;
	size_t max_chunks;
	size_t max_chunk_size;
	size_t size;
	size_t chunks;
	size_t bytes;

	cl_mem d_data;
	cl_mem d_indices;
	cl_mem d_sizes;
	cl_mem d_results;
	cl_mem d_results2;
	cl_mem d_prefixsum;
	cl_mem d_results_comp;
	cl_mem d_results2_comp;

	cl_mem p_data;		/* pinned twins: the h_* arrays ARE pinned, */
	cl_mem p_indices;	/* these stay NULL                          */
	cl_mem p_sizes;
	cl_mem p_results;
	cl_mem p_results2;
	cl_mem p_prefixsum;
	cl_mem p_results_comp;
	cl_mem p_results2_comp;

	cl_mem *ScanPartialSums;	/* unused: scan scratch is in 'ws'  */
	unsigned int ScanPartialSums_size;

	struct clconf *cl;

	/* additions of this library */
	void *ws;		/* device scratch of the scan pipeline      */
	size_t ws_bytes;
	int compact;		/* 1: process_results reads the compact
				 * planes (the reference's COMPACT_RESULTS
				 * build, databuf.c:16); 0: bucket planes   */
	int scanned;		/* planes on the device are current         */
	void *pack_text;	/* device copy of a padded chunk list packed */
	size_t pack_text_cap;	/* into one stream (ocl_aho_match), and the  */
	int *pack_starts;	/* packed start of each chunk: device array  */
	int *h_pack_starts;	/* and its pinned host twin                  */
	size_t pack_starts_cap;
};

struct databuf *databuf_new(size_t max_chunks, size_t max_chunk_size,
    int max_results, int mapped, struct clconf *);	/* databuf.h:79 */
int databuf_add_fd(struct databuf *, int fd, int id, size_t *rd_bytes);
int databuf_add_fp(struct databuf *, FILE *, int id, int aligned,
    size_t *rd_bytes, size_t *rd_lines);
int databuf_add_chunk(struct databuf *, char *chunk, size_t len, int id,
    char aligned);					/* databuf.c:487 */
void databuf_reset(struct databuf *);
void databuf_clear(struct databuf *);
void databuf_copy_host_to_device(struct databuf *, cl_command_queue);
void databuf_copy_device_to_host(struct databuf *, cl_command_queue);
int databuf_process_results(struct databuf *,
    int (*cb)(int file_idx, int patrn_idx, int chunk_idx, int offset,
    void *uarg), void *uarg);
void databuf_free(struct databuf *, int mapped, cl_command_queue);

/* ---- ocl_aho_match.h:12-30 --------------------------------------------- */
void ocl_aho_match_init(struct clconf *);
void ocl_aho_match_close(struct clconf *);
/* scans db (device copy) and fills the bucket planes AND the compact
 * planes; blocks until done like the reference's clFinish
 * (ocl_aho_match.c:125-130).  'stream' is accepted and ignored (:83-90). */
void ocl_aho_match(struct clconf *, struct databuf *, acsm_t *,
    size_t local_ws, int stream);

/* ---- ocl_prefix_sum.h:12-22, ocl_compact_array.h:12-22 ----------------- */
void ocl_prefix_sum_init(struct clconf *);
void ocl_prefix_sum_close(struct clconf *);
void ocl_prefix_sum(struct clconf *, struct databuf *, unsigned int n);
void ocl_compact_array_init(struct clconf *);
void ocl_compact_array_close(struct clconf *);
void ocl_compact_array(struct clconf *, struct databuf *, size_t local_ws);

/* ---- ocl_bitonic_sort.h:13-18 ------------------------------------------ */
int ocl_bitonic_sort_init(struct clconf *);
int ocl_bitonic_sort_close(struct clconf *);
int ocl_bitonic_sort(struct clconf *, cl_mem key_dst, cl_mem val_dst,
    cl_mem key_src, cl_mem val_src, unsigned int batch, unsigned int len,
    unsigned int dir);

#ifdef __cplusplus
}
#endif
#endif /* ACMATCH_H_ */
