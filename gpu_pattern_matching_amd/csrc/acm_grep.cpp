// acm_grep -- MI355X-native counterpart of the reference's ocl_aho_grep
// (streaming feeder, CLI + stdout contract,
// directory traversal / multi-file packing).
//
// Same flags, same -v line and STATS block as ocl_aho_grep.c:151-204,
// :272-308, :615-631, so apps/sentiment_analysis.py-style consumers keep
// working.  What is different underneath:
//
//   * each worker (-w) owns one HIP stream and TWO staging buffers: while the
//     GPU copies in and scans buffer A (hipMemcpyAsync from pinned memory ->
//     acm_scan_async -> bucket planes -> async copy back), the worker thread
//     is already read(2)-ing the next bytes into buffer B.  The reference
//     reads, copies and scans strictly one after the other
//     (ocl_aho_grep.c:68-139, blocking CL_TRUE copies + clFinish);
//   * one DFA per device, shared by all workers (the reference uploads a
//     private copy per worker, ocl_worker.c:66,149-153 -- quirk Q12);
//   * matches are those of a serial scan, the state is
//     carried from buffer to buffer of a worker like db->last_state.
//
// Files are dealt to workers as in the reference: worker i takes files
// i, i + w, i + 2w, ... (ocl_aho_grep.c:47,87).
#include <dirent.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "acmatch.h"

namespace {

volatile sig_atomic_t g_terminate = 0;
void on_sigint(int) { g_terminate = 1; }

double now_us()
{
	struct timespec tp;
	clock_gettime(CLOCK_MONOTONIC, &tp);  // utils.c:60-68
	return tp.tv_sec * 1e6 + tp.tv_nsec / 1e3;
}

[[noreturn]] void usage()
{
	printf("\nUsage:\n"
	       "    acm_grep -f file -p file -B chunk_size -D devpos\n"
	       "             -G global_ws -L local_ws [-m max]\n"
	       "             [-w cpu_threads] [-R max] [-tvxFM]\n"
	       "    acm_grep -h\n\n"
	       "Options (those of ocl_aho_grep):\n"
	       "  -f file        input: a file, a directory, or comma-separated files\n"
	       "  -p file        patterns, one per line (plain, \"quoted\", or 'ID pattern')\n"
	       "  -F             keep processing data appended to the inputs (e.g. a FIFO)\n"
	       "  -B chunk_size  chunk size in bytes (result buckets are per chunk)\n"
	       "  -D devpos      HIP device ordinal, or a list (0,1,2,...): worker i runs on entry i mod\n"
	       "                 the list length, each device holds its own copy of the automaton\n"
	       "  -G global_ws   chunks per buffer: buffer = global_ws * chunk_size bytes\n"
	       "  -L local_ws    accepted for compatibility; the launch shape is the library's\n"
	       "  -m max         use at most max bytes of every pattern\n"
	       "  -w threads     feeder threads (default 2)\n"
	       "  -R max         result cells per chunk incl. the counter cell (default 16)\n"
	       "  -v             print every match\n"
	       "  -t             text mode: one chunk per line\n"
	       "  -x             patterns are printable hex\n"
	       "  -M             accepted for compatibility (mapped buffers)\n"
	       "  -A             report every pattern that ends at an offset, not only the one the\n"
	       "                 reference reports (extension; off by default)\n"
	       "  -h             this help\n");
	exit(EXIT_FAILURE);
}

bool is_dir(const std::string &p)
{
	struct stat st;
	return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

bool is_readable_input(const std::string &p)
{
	struct stat st;
	return stat(p.c_str(), &st) == 0 && (S_ISREG(st.st_mode) || S_ISFIFO(st.st_mode));
}

// file_traverse.c:107-166: regular files directly under dir (not recursive)
std::vector<std::string> regular_files_in(std::string dir)
{
	std::vector<std::string> out;
	if (!dir.empty() && dir.back() == '/')
		dir.pop_back();
	DIR *d = opendir(dir.c_str());
	if (!d)
		return out;
	while (struct dirent *e = readdir(d)) {
		if (e->d_type == DT_DIR)
			continue;
		std::string f = dir + "/" + e->d_name;
		struct stat st;
		if (stat(f.c_str(), &st) == 0 && S_ISREG(st.st_mode))
			out.push_back(f);
	}
	closedir(d);
	return out;
}

struct Config {
	std::string pat_path, data_path;
	int dev = -1, hex = 0, verbose = 0, text_mode = 0, follow = 0, threads = 2, all_patterns = 0;
	std::vector<int> devs;   // -D 0,1,...: worker i runs on devs[i % devs.size()] (the reference has one -D)
	int max_results = MAX_RESULTS, pat_limit = -1;
	long global_ws = -1, local_ws = -1, chunk = -1;
};

struct Shared {
	Config cfg;
	std::vector<acm_dfa *> dfas;   // one per entry of cfg.devs
	std::vector<std::string> files;
	std::vector<int> fds;
	std::vector<std::string> pat_bytes;   // for the -v line
	std::vector<int> pat_iid;
	pthread_mutex_t print_lock = PTHREAD_MUTEX_INITIALIZER;
	pthread_barrier_t ready;   // workers have their buffers; the clock starts (the reference
	                           // also allocates in ocl_worker_ctx_init, before start_time)
};

constexpr size_t kAllFactor = 4;   // -A: records the expanded planes hold per text byte

struct Buffer {   // one of the two staging buffers of a worker
	unsigned char *h_data = nullptr;
	int32_t *h_indices = nullptr, *h_sizes = nullptr, *file_ids = nullptr;
	int32_t *h_results = nullptr, *h_results2 = nullptr;
	void *d_data = nullptr, *d_indices = nullptr, *d_sizes = nullptr, *d_starts = nullptr;
	void *d_results = nullptr, *d_results2 = nullptr, *d_pat = nullptr, *d_off = nullptr;
	void *d_pat_all = nullptr, *d_off_all = nullptr, *d_expand_ws = nullptr;   // -A only
	size_t all_cap = 0;          // cells of the -A planes
	int32_t *h_all_count = nullptr;   // pinned: records the expansion produced
	size_t expand_ws_bytes = 0;
	void *d_packed = nullptr;
	size_t chunks = 0, bytes = 0;
	std::vector<int32_t> starts;
	void *done = nullptr;        // event behind the buffer's last copy: collect() waits for this buffer, not for the stream
	size_t scan_cap = 0;         // capacity its scan's planes were given (the next buffer reads its final state from them)
};

struct Worker {
	acm_dfa *dfa = nullptr;   // the copy on this worker's device
	int dev = 0;
	Shared *sh = nullptr;
	int id = 0;
	pthread_t thread{};
	void *stream = nullptr, *ws = nullptr;
	size_t ws_bytes = 0;
	Buffer buf[2];
	long last_state = 0;
	size_t matches = 0, reported = 0, bytes = 0, lines = 0, rounds = 0;
};

void die_acm(const char *what)
{
	fprintf(stderr, "ERROR: %s: %s\n", what, acm_last_error());
	exit(1);
}

#define CK(call) do { if ((call) != ACM_OK) die_acm(#call); } while (0)

void buffer_alloc(Buffer &b, const Config &c)
{
	const size_t size = (size_t)c.global_ws * c.chunk, G = (size_t)c.global_ws;
	const size_t plane = ((size_t)c.max_results * G + 1) * 4;
	CK(acm_rt_host_alloc((void **)&b.h_data, size + 32));
	CK(acm_rt_host_alloc((void **)&b.h_indices, (G + 1) * 4));
	CK(acm_rt_host_alloc((void **)&b.h_sizes, (G + 1) * 4));
	CK(acm_rt_host_alloc((void **)&b.h_results, plane));
	CK(acm_rt_host_alloc((void **)&b.h_results2, plane));
	b.file_ids = (int32_t *)calloc(G + 1, 4);
	CK(acm_rt_malloc(&b.d_data, size + 32));
	CK(acm_rt_malloc(&b.d_packed, size + 32));
	CK(acm_rt_malloc(&b.d_indices, (G + 1) * 4));
	CK(acm_rt_malloc(&b.d_sizes, (G + 1) * 4));
	CK(acm_rt_malloc(&b.d_starts, (G + 2) * 4));
	CK(acm_rt_malloc(&b.d_results, plane));
	CK(acm_rt_malloc(&b.d_results2, plane));
	CK(acm_rt_malloc(&b.d_pat, (size + 2) * 4));
	CK(acm_rt_malloc(&b.d_off, (size + 2) * 4));
	if (c.all_patterns) {
		b.expand_ws_bytes = acm_expand_workspace_bytes(size);
		// every pattern of every final state's list: more records than text bytes when patterns nest
		// (aaa, aaaa, aaaaa over a run of a's); the planes hold kAllFactor per byte, beyond that the
		// buffer is reported as an error (collect), never overrun
		b.all_cap = size * kAllFactor + 2;
		CK(acm_rt_host_alloc((void **)&b.h_all_count, 64));
		*b.h_all_count = 0;
		CK(acm_rt_malloc(&b.d_pat_all, b.all_cap * 4));
		CK(acm_rt_malloc(&b.d_off_all, b.all_cap * 4));
		CK(acm_rt_malloc(&b.d_expand_ws, b.expand_ws_bytes));
	}
}

// binary mode: fixed chunks, a short tail chunk per file (databuf.c:326-407)
// returns bytes read (0 = nothing available right now)
size_t fill_binary(Buffer &b, const Config &c, int fd, int file_id)
{
	const size_t G = (size_t)c.global_ws, B = (size_t)c.chunk;
	if (b.chunks >= G)
		return 0;
	const ssize_t got = read(fd, b.h_data + b.chunks * B, (G - b.chunks) * B);
	if (got <= 0)
		return 0;
	size_t left = (size_t)got;
	while (left) {
		const size_t take = std::min(left, B);
		b.h_indices[b.chunks] = (int32_t)(b.chunks * B);
		b.h_sizes[b.chunks] = (int32_t)take;
		b.file_ids[b.chunks] = file_id;
		b.chunks++;
		left -= take;
	}
	b.bytes = b.chunks * B;
	return (size_t)got;
}

// text mode: one chunk per line, 16-byte aligned, gaps zeroed (databuf.c:412-481)
size_t fill_text(Buffer &b, const Config &c, FILE *fp, int file_id, size_t *lines)
{
	const size_t G = (size_t)c.global_ws, B = (size_t)c.chunk, size = G * B;
	size_t total = 0;
	while (b.chunks < G && b.bytes < size) {
		const size_t room = std::min(size - b.bytes, B);
		if (room < 2)
			break;
		char *dst = (char *)b.h_data + b.bytes;
		if (!fgets(dst, (int)room, fp))
			break;
		const size_t len = strnlen(dst, room);
		if (len && dst[len - 1] == '\n')
			(*lines)++;
		b.h_indices[b.chunks] = (int32_t)b.bytes;
		b.h_sizes[b.chunks] = (int32_t)len;
		b.file_ids[b.chunks] = file_id;
		b.chunks++;
		const size_t adv = std::min((len + 15) & ~(size_t)15, size - b.bytes);
		memset(dst + len, 0, adv - len);
		b.bytes += adv;
		total += len;
	}
	return total;
}

// enqueue copy-in, scan, bucket planes and copy-back of one buffer; no sync.  prev: the worker's buffer
// in front of this one if its results have not been collected yet -- the scan then starts in the state
// that buffer's scan ended in, read from its planes ON THE DEVICE (acm_scan_batch.d_init_plane; the
// reference carries it through the host, databuf.c:622): the GPU goes on with this buffer while the
// host walks the previous one's results.
void submit(Worker &w, Buffer &b, const Buffer *prev)
{
	const Config &c = w.sh->cfg;
	const int chunks = (int)b.chunks;
	void *s = w.stream;
	size_t stream_len = 0;
	bool packed = true;
	b.starts.resize((size_t)chunks + 1);
	for (int i = 0; i < chunks; i++) {
		if ((size_t)b.h_indices[i] != stream_len)
			packed = false;
		b.starts[i] = (int32_t)stream_len;
		stream_len += (size_t)b.h_sizes[i];
	}
	b.starts[chunks] = (int32_t)stream_len;
	CK(acm_rt_memcpy_h2d(b.d_data, b.h_data, (b.bytes + 15) & ~(size_t)15, s));
	CK(acm_rt_memcpy_h2d(b.d_indices, b.h_indices, (size_t)chunks * 4, s));
	CK(acm_rt_memcpy_h2d(b.d_sizes, b.h_sizes, (size_t)chunks * 4, s));
	size_t cap = (size_t)c.global_ws * c.chunk + 2;
	const void *text = b.d_data;
	if (!packed) {   // padded chunk list: scan the chunks' bytes back to back
		CK(acm_rt_memcpy_h2d(b.d_starts, b.starts.data(), ((size_t)chunks + 1) * 4, s));
		CK(acm_pack_chunks(b.d_packed, b.d_data, (const int32_t *)b.d_indices, (const int32_t *)b.d_sizes,
		    (const int32_t *)b.d_starts, chunks, s));
		text = b.d_packed;
	}
	int32_t *pat = (int32_t *)b.d_pat, *off = (int32_t *)b.d_off;
	acm_scan_batch sb;
	memset(&sb, 0, sizeof(sb));
	sb.d_text = text;
	sb.n = stream_len;
	sb.init_state = w.last_state;
	if (prev) {
		sb.init_state = 0;
		sb.d_init_plane = (const int32_t *)prev->d_pat;
		sb.init_plane_capacity = prev->scan_cap;
	}
	sb.d_workspace = w.ws;
	sb.workspace_bytes = w.ws_bytes;
	sb.d_pat_plane = pat;
	sb.d_off_plane = off;
	sb.plane_capacity = cap;
	sb.stream = s;
	b.scan_cap = cap;
	if (!c.all_patterns) {
		CK(acm_scan_batch_async(w.dfa, &sb));
	} else {   // final states instead of head patterns, then every pattern of each state's match list
		sb.report = ACM_REPORT_STATE;
		CK(acm_scan_batch_async(w.dfa, &sb));
		CK(acm_expand_matches_async(w.dfa, pat, off, cap - 2, (int32_t *)b.d_pat_all, (int32_t *)b.d_off_all,
		    b.all_cap, b.d_expand_ws, b.expand_ws_bytes, s));
		pat = (int32_t *)b.d_pat_all;
		off = (int32_t *)b.d_off_all;
		cap = b.all_cap;
		CK(acm_rt_memcpy_d2h(b.h_all_count, pat, 4, s));
	}
	if (!packed)
		CK(acm_remap_offsets(off, cap - 2, (const int32_t *)b.d_indices,
		    (const int32_t *)b.d_starts, chunks, s));
	CK(acm_bucketize(pat, off, (const int32_t *)b.d_indices,
	    (const int32_t *)b.d_sizes, chunks, c.max_results, (int32_t *)b.d_results, (int32_t *)b.d_results2, cap, s));
	const size_t cells = (size_t)c.max_results * chunks + 1;
	CK(acm_rt_memcpy_d2h(b.h_results, b.d_results, cells * 4, s));
	CK(acm_rt_memcpy_d2h(b.h_results2, b.d_results2, cells * 4, s));
	if (!b.done)
		CK(acm_rt_event_create(&b.done));
	CK(acm_rt_event_record(b.done, s));
}

// wait for the buffer, walk the bucket planes (databuf.c:747-782), print -v lines
void collect(Worker &w, Buffer &b)
{
	const Config &c = w.sh->cfg;
	CK(acm_rt_event_sync(b.done));   // (this buffer's copies; the next buffer's scan may still be running)
	if (c.all_patterns && b.h_all_count && (size_t)*b.h_all_count > b.all_cap - 2) {
		fprintf(stderr, "ERROR: -A produced %d records for one buffer, the planes hold %zu; use a smaller -G/-B\n",
		    *b.h_all_count, b.all_cap - 2);
		exit(1);
	}
	const size_t chunks = b.chunks;
	const int R = c.max_results;
	w.last_state = b.h_results[chunks * R];
	for (size_t i = 0; i < chunks; i++) {
		const int n = b.h_results[i];
		w.matches += (size_t)n;
		for (int j = 0; j < n && j < R - 1; j++) {
			const int p_idx = b.h_results[(size_t)(j + 1) * chunks + i];
			const int off = b.h_results2[(size_t)(j + 1) * chunks + i] + 1;  // end + 1 (databuf.c:771)
			w.reported++;
			if (!c.verbose)
				continue;
			pthread_mutex_lock(&w.sh->print_lock);
			printf("Pattern %d ('%s') found in file '%s' at offset %d [relative: %d]\n",
			    w.sh->pat_iid[p_idx], w.sh->pat_bytes[p_idx].c_str(),
			    w.sh->files[b.file_ids[i]].c_str(), off, off - b.h_indices[i]);
			if (c.text_mode) {   // the matching line
				fwrite(b.h_data + b.h_indices[i], 1, (size_t)b.h_sizes[i], stdout);
			} else {             // some context around the match, up to a newline
				printf(" ... ");
				const int plen = (int)w.sh->pat_bytes[p_idx].size();
				const int lo = std::max(0, off - plen - 10);
				for (int k = lo; k < off + 10 && (size_t)k < b.bytes; k++) {
					if (b.h_data[k] == '\n')
						break;
					putchar(b.h_data[k]);
				}
				printf(" ... \n");
			}
			pthread_mutex_unlock(&w.sh->print_lock);
		}
	}
	b.chunks = 0;
	b.bytes = 0;
	w.rounds++;
}

void *worker_main(void *arg)
{
	Worker &w = *(Worker *)arg;
	Shared &sh = *w.sh;
	const Config &c = sh.cfg;
	CK(acm_rt_set_device(w.dev));
	CK(acm_rt_stream_create(&w.stream));
	w.ws_bytes = acm_scan_workspace_bytes(w.dfa, (size_t)c.global_ws * c.chunk);
	CK(acm_rt_malloc(&w.ws, w.ws_bytes));
	buffer_alloc(w.buf[0], c);
	buffer_alloc(w.buf[1], c);
	pthread_barrier_wait(&sh.ready);

	const int nfiles = (int)sh.files.size();
	int cur = w.id, filling = 0;
	bool in_flight = false;
	// text mode: one FILE per input, opened once (follow mode comes back to the same inputs
	// every millisecond: a fresh fdopen per visit would leak a stdio buffer each time)
	std::vector<FILE *> fps(c.text_mode ? (size_t)nfiles : 0, nullptr);
	auto stream_of = [&](int f) -> FILE * {
		if (!fps[(size_t)f])
			fps[(size_t)f] = fdopen(sh.fds[f], "r");
		else
			clearerr(fps[(size_t)f]);   // past an EOF seen earlier: data may have been appended
		return fps[(size_t)f];
	};
	FILE *fp = nullptr;
	if (cur < nfiles && c.text_mode)
		fp = stream_of(cur);
	const size_t G = (size_t)c.global_ws, size = G * (size_t)c.chunk;

	while (cur < nfiles) {
		Buffer &b = w.buf[filling];
		size_t got, lines = 0;
		if (c.text_mode)
			got = fill_text(b, c, fp, cur, &lines);
		else
			got = fill_binary(b, c, sh.fds[cur], cur);
		w.bytes += got;
		w.lines += lines;
		const bool full = b.chunks >= G || b.bytes + 2 > size;
		bool file_done = (got == 0) && !full;
		if (file_done) {   // current file exhausted: next one of this worker
			if (!c.follow) {
				if (fp) {
					fclose(fp);
					fps[(size_t)cur] = nullptr;
					fp = nullptr;
				} else {
					close(sh.fds[cur]);
				}
			}
			cur += c.threads;
			if (cur >= nfiles && c.follow && !g_terminate) {
				cur = w.id;   // keep polling the inputs (ocl_aho_grep.c:96-99)
				usleep(1000);
			}
			if (cur < nfiles && c.text_mode)
				fp = stream_of(cur);
		}
		const bool last = cur >= nfiles || g_terminate;
		if (b.chunks > 0 && (full || last || (c.follow && file_done))) {
			// the scan starts in the state the previous buffer ended in: taken from that buffer's planes on
			// the device, so this one is enqueued BEFORE the host waits for the previous one and walks its results
			submit(w, b, in_flight ? &w.buf[filling ^ 1] : nullptr);   // GPU works on b while we read into the other buffer
			if (in_flight)
				collect(w, w.buf[filling ^ 1]);
			in_flight = true;
			filling ^= 1;
		}
		if (g_terminate)
			break;
	}
	if (in_flight)
		collect(w, w.buf[filling ^ 1]);
	return nullptr;
}

}  // namespace

int main(int argc, char **argv)
{
	// one hardware queue per worker stream (HIP's default is 4 per process; streams that share a
	// queue serialise).  Has to be in the environment before the first HIP call.
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	Shared sh;
	Config &c = sh.cfg;
	int opt;
	while ((opt = getopt(argc, argv, "f:m:p:tw:vxB:D:FG:L:R:MhA")) != -1) {   // ocl_aho_grep.c:411 + A
		switch (opt) {
		case 'f': c.data_path = optarg; break;
		case 'm': c.pat_limit = atoi(optarg); break;
		case 'p': c.pat_path = optarg; break;
		case 't': c.text_mode = 1; break;
		case 'w': c.threads = atoi(optarg); break;
		case 'v': c.verbose = 1; break;
		case 'x': c.hex = 1; break;
		case 'B': c.chunk = atol(optarg); break;
		case 'D':   // one device, or a comma-separated list: the workers are dealt over it
			c.devs.clear();
			for (const char *q = optarg; *q;) {
				c.devs.push_back(atoi(q));
				while (*q && *q != ',')
					q++;
				if (*q == ',')
					q++;
			}
			c.dev = c.devs.empty() ? -1 : c.devs[0];
			break;
		case 'F': c.follow = 1; break;
		case 'G': c.global_ws = atol(optarg); break;
		case 'L': c.local_ws = atol(optarg); break;
		case 'R': c.max_results = atoi(optarg); break;
		case 'M': break;
		case 'A': c.all_patterns = 1; break;
		default: usage();
		}
	}
	int err = 0;   // check_args, ocl_aho_grep.c:210-266
	if (c.pat_path.empty()) { printf("ERROR: No pattern file\n"); err++; }
	else if (access(c.pat_path.c_str(), R_OK) != 0) {
		printf("ERROR: File '%s' does not exist\n", c.pat_path.c_str()); err++;
	}
	if (c.data_path.empty()) { printf("ERROR: No data file\n"); err++; }
	if (c.dev == -1) { printf("ERROR: No device position\n"); err++; }
	if (c.global_ws == -1) { printf("ERROR: No global work size\n"); err++; }
	if (c.local_ws == -1) { printf("ERROR: No local work size\n"); err++; }
	if (c.chunk == -1) { printf("ERROR: No maximum chunk size\n"); err++; }
	if (c.threads <= 0) { printf("ERROR: The thread number must be greater than 0\n"); err++; }
	if (c.pat_limit != -1 && c.pat_limit <= 0) { printf("ERROR: The pattern size limit should be >= 1\n"); err++; }
	if (c.pat_limit >= 4096) { printf("ERROR: The pattern size limit should be <= 4095\n"); err++; }
	if (c.max_results <= 0) { printf("ERROR: The maximum result cells should be >= 1\n"); err++; }
	if (err)
		usage();
	auto align16 = [](long &v, const char *what) {   // align_parameters, ocl_aho_grep.c:316-346
		if (v % 16) {
			printf("WARNING: %s '%ld' is not 16B aligned. ", what, v);
			v = (v + 15) & ~15L;
			printf("Will use '%ld' instead\n", v);
		}
	};
	align16(c.local_ws, "local work size");
	align16(c.global_ws, "global work size");
	align16(c.chunk, "max chunk size");
	printf("Local Work Size:  %ld\nGlobal Work Size: %ld\nMax Chunk Size:   %ld\n\n", c.local_ws,
	    c.global_ws, c.chunk);

	struct rlimit rl;
	if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < rl.rlim_max) {
		rl.rlim_cur = rl.rlim_max;
		setrlimit(RLIMIT_NOFILE, &rl);
	}

	// inputs: a directory, one file, or comma-separated files
	std::vector<std::string> names;
	if (is_dir(c.data_path)) {
		names = regular_files_in(c.data_path);
	} else {
		size_t a = 0;
		while (a <= c.data_path.size()) {
			size_t b = c.data_path.find(',', a);
			if (b == std::string::npos)
				b = c.data_path.size();
			if (b > a)
				names.push_back(c.data_path.substr(a, b - a));
			a = b + 1;
		}
	}
	for (auto &f : names) {
		if (!is_readable_input(f))
			continue;
		int fd = open(f.c_str(), O_RDONLY);
		if (fd == -1) {
			fprintf(stderr, "ERROR: could not open '%s'\n\n", f.c_str());
			return 1;
		}
		sh.files.push_back(f);
		sh.fds.push_back(fd);
	}
	if (sh.files.empty()) {
		fprintf(stderr, "ERROR: Could not open input file(s) for reading.\n\n");
		return 1;
	}

	// one automaton, one copy per device, shared by the workers of that device
	acm_automaton *aut = acm_automaton_new();
	if (acm_automaton_load_file(aut, c.pat_path.c_str(), c.hex, c.pat_limit) < 0) {
		fprintf(stderr, "ERROR: init_ocl_worker_ctx\n%s\n", acm_last_error());
		return 1;
	}
	CK(acm_automaton_compile(aut));
	const int states = acm_automaton_num_states(aut);
	const int np = acm_automaton_num_patterns(aut);
	for (int i = 0; i < np; i++) {
		int iid = 0, n = 0;
		const unsigned char *bytes = nullptr;
		acm_automaton_pattern(aut, i, &iid, &n, &bytes, nullptr);
		sh.pat_iid.push_back(iid);
		sh.pat_bytes.emplace_back((const char *)bytes, (size_t)n);
	}
	for (int dev : c.devs) {
		acm_dfa *dfa = nullptr;
		if (acm_dfa_upload(aut, dev, &dfa) != ACM_OK) {
			fprintf(stderr, "invalid dev pos\n%s\n", acm_last_error());
			return 1;
		}
		sh.dfas.push_back(dfa);
	}
	const size_t automaton_bytes = acm_dfa_device_bytes(sh.dfas[0]);
	acm_automaton_free(aut);

	signal(SIGINT, on_sigint);
	std::vector<Worker> workers((size_t)c.threads);
	pthread_barrier_init(&sh.ready, nullptr, (unsigned)c.threads + 1);
	for (int i = 0; i < c.threads; i++) {
		workers[i].sh = &sh;
		workers[i].id = i;
		workers[i].dev = c.devs[(size_t)i % c.devs.size()];
		workers[i].dfa = sh.dfas[(size_t)i % c.devs.size()];
		if (pthread_create(&workers[i].thread, nullptr, worker_main, &workers[i]) != 0) {
			fprintf(stderr, "ERROR: creating thread: %d\n\n", i);
			return 1;
		}
	}
	pthread_barrier_wait(&sh.ready);
	const double t0 = now_us();
	size_t matches = 0, reported = 0, bytes = 0, lines = 0, rounds = 0;
	for (auto &w : workers) {
		pthread_join(w.thread, nullptr);
		matches += w.matches;
		reported += w.reported;
		bytes += w.bytes;
		lines += w.lines;
		rounds += w.rounds;
	}
	const double secs = (now_us() - t0) / 1e6;
	printf("-------------- STATS --------------\n");   // ocl_aho_grep.c:615-631
	printf("Matches:             %lu\n", (unsigned long)matches);
	printf("Matches reported:    %lu\n", (unsigned long)reported);
	printf("Time (secs):         %.5f\n", secs);
	printf("Automaton states:    %d\n", states);
	printf("Automaton size (MB): %.3f\n", (double)automaton_bytes / 1048576);
	printf("Processed bytes:     %lu\n", (unsigned long)bytes);
	if (lines)
		printf("Processed lines:     %lu\n", (unsigned long)lines);
	printf("Processed files:     %d\n", (int)sh.files.size());
	printf("Kernel launches:     %d\n", (int)rounds);
	printf("Throughput (Mbps):   %.3f\n", ((double)(bytes * 8) / 1048576) / secs);
	printf("-----------------------------------\n\n");
	for (acm_dfa *dfa : sh.dfas)
		acm_dfa_release(dfa);
	return 0;
}
