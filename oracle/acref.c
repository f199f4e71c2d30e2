/*
 * oracle/acref.c -- CPU restatement of the reference's Aho-Corasick path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gpu_pattern_matching_amd/ links,
 * loads or calls this file; it is the checker used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.
 *
 * What is restated (file:line into the reference tree):
 *   - pattern list + LIFO insertion .............. acsmx.c:514-546
 *   - trie construction / state numbering ........ acsmx.c:318-349, 552-585
 *   - fail links + match-list inheritance order .. acsmx.c:355-438
 *   - DFA completion ............................. acsmx.c:444-486
 *   - serialised table  [state][2][256] int32 .... acsmx.c:600-671
 *   - patterns table with shared-final chains .... acsmx.c:677-735
 *   - pattern-file parser (plain/hex/categorical). ocl_worker.c:74-145,
 *                                                  utils.c:18-54
 *   - serial scan over the table (what the removed acsmSearch would do;
 *     semantics of one work-item of ahomatch.cl:50-77 run over the whole
 *     buffer) ..................................... SURVEY.md App. B.1
 *   - reference kernel chunk semantics (compat) .. ahomatch.cl:1-165
 *   - exclusive prefix sum of per-chunk counts ... ocl_prefix_sum.c:164-221
 *   - bucket -> dense compaction ................. compactarray.cl:40-68
 *   - key/value bitonic sort ..................... BitonicSort.cl:19-45
 *   - bucket walk + callback offsets ............. databuf.c:747-782
 *
 * Pinning: every function here is checked against /root/reference/acsmx.c
 * compiled as-is (oracle/_ref, see oracle/Makefile + ref_harness.c) and
 * against the known-answer vectors of SURVEY.md App. D; digests produced by
 * the compiled reference are committed under tests/golden/.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_ALPHA 256
#define ORC_FAIL (-1)
#define ORC_MAX_PAT_LINE 4096 /* utils.h:14 MAX_PAT_SIZE */

typedef struct {
	unsigned char *bytes;
	int n;
	int iid;
	int index; /* forward insertion ordinal, acsmx.c:535 */
} orc_pat_t;

/* match-list node; lists are singly linked through 'next' (index into nodes) */
typedef struct {
	int pat; /* pattern index */
	int next; /* -1 terminates */
} orc_mnode_t;

typedef struct {
	orc_pat_t *pats; /* in insertion order */
	int npats, cap_pats;
	int max_pattern_len;

	int max_states; /* 1 + sum of lengths, acsmx.c:560-562 */
	int num_states; /* highest state id during build; +1 after gen_table */
	int32_t *next; /* [max_states][256] */
	int32_t *fail; /* [max_states] */
	int32_t *mhead; /* [max_states] head of match list or -1 */
	orc_mnode_t *mnodes;
	size_t n_mnodes, cap_mnodes;

	int32_t *table; /* [num_states][2][256], reference device format */
	int table_states;
} orc_t;

/* ------------------------------------------------------------------ build */

orc_t *
orc_new(void)
{
	return (orc_t *)calloc(1, sizeof(orc_t));
}

void
orc_free(orc_t *o)
{
	int i;
	if (!o)
		return;
	for (i = 0; i < o->npats; i++)
		free(o->pats[i].bytes);
	free(o->pats);
	free(o->next);
	free(o->fail);
	free(o->mhead);
	free(o->mnodes);
	free(o->table);
	free(o);
}

/* acsmx.c:514-546 -- index = forward ordinal; the list is walked newest
 * first at compile time, which we reproduce by iterating pats[] backwards. */
void
orc_add_pattern(orc_t *o, const unsigned char *pat, int n, int iid)
{
	orc_pat_t *p;
	if (o->npats == o->cap_pats) {
		o->cap_pats = o->cap_pats ? o->cap_pats * 2 : 1024;
		o->pats = (orc_pat_t *)realloc(o->pats,
		    (size_t)o->cap_pats * sizeof(orc_pat_t));
	}
	p = &o->pats[o->npats];
	p->bytes = (unsigned char *)malloc(n > 0 ? (size_t)n : 1);
	memcpy(p->bytes, pat, (size_t)n);
	p->n = n;
	p->iid = iid;
	p->index = o->npats;
	o->npats++;
	if (n > o->max_pattern_len)
		o->max_pattern_len = n;
}

static int
orc_mnode_new(orc_t *o, int pat, int next)
{
	if (o->n_mnodes == o->cap_mnodes) {
		o->cap_mnodes = o->cap_mnodes ? o->cap_mnodes * 2 : 4096;
		o->mnodes = (orc_mnode_t *)realloc(o->mnodes,
		    o->cap_mnodes * sizeof(orc_mnode_t));
	}
	o->mnodes[o->n_mnodes].pat = pat;
	o->mnodes[o->n_mnodes].next = next;
	return (int)o->n_mnodes++;
}

/* acsmx.c:318-349 */
static void
orc_add_pattern_states(orc_t *o, const orc_pat_t *p)
{
	int state = 0, n = p->n, nx;
	const unsigned char *c = p->bytes;

	for (; n > 0; c++, n--) {
		nx = o->next[(size_t)state * ORC_ALPHA + *c];
		if (nx == ORC_FAIL)
			break;
		state = nx;
	}
	for (; n > 0; c++, n--) {
		o->num_states++;
		o->next[(size_t)state * ORC_ALPHA + *c] = o->num_states;
		state = o->num_states;
	}
	/* add_match_list_entry: push at the front (acsmx.c:299-312) */
	o->mhead[state] = orc_mnode_new(o, p->index, o->mhead[state]);
}

void
orc_compile(orc_t *o)
{
	int i, k, r, s, fs, nx, m;
	int *queue;
	size_t qh, qt;

	o->max_states = 1;
	for (i = 0; i < o->npats; i++)
		o->max_states += o->pats[i].n;
	o->next = (int32_t *)malloc((size_t)o->max_states * ORC_ALPHA *
	    sizeof(int32_t));
	o->fail = (int32_t *)calloc((size_t)o->max_states, sizeof(int32_t));
	o->mhead = (int32_t *)malloc((size_t)o->max_states * sizeof(int32_t));
	if (!o->next || !o->fail || !o->mhead) {
		fprintf(stderr, "oracle: out of memory\n");
		exit(1);
	}
	for (k = 0; k < o->max_states; k++)
		o->mhead[k] = -1;
	memset(o->next, 0xff, (size_t)o->max_states * ORC_ALPHA *
	    sizeof(int32_t)); /* all ORC_FAIL (-1) */
	o->num_states = 0;

	/* newest pattern first (acsmx.c:536-538 prepend, :579-580 walk) */
	for (i = o->npats - 1; i >= 0; i--)
		orc_add_pattern_states(o, &o->pats[i]);

	for (i = 0; i < ORC_ALPHA; i++)
		if (o->next[i] == ORC_FAIL)
			o->next[i] = 0;

	queue = (int *)malloc((size_t)o->max_states * sizeof(int));

	/* build_NFA, acsmx.c:355-438 */
	qh = qt = 0;
	for (i = 0; i < ORC_ALPHA; i++) {
		s = o->next[i];
		if (s) {
			queue[qt++] = s;
			o->fail[s] = 0;
		}
	}
	while (qh < qt) {
		r = queue[qh++];
		for (i = 0; i < ORC_ALPHA; i++) {
			s = o->next[(size_t)r * ORC_ALPHA + i];
			if (s == ORC_FAIL)
				continue;
			queue[qt++] = s;
			fs = o->fail[r];
			while ((nx = o->next[(size_t)fs * ORC_ALPHA + i]) ==
			    ORC_FAIL)
				fs = o->fail[fs];
			o->fail[s] = nx;
			/* copy nx's list, each copy pushed at the FRONT of s's
			 * list, walking nx's list head to tail (:417-429) */
			for (m = o->mhead[nx]; m != -1; m = o->mnodes[m].next)
				o->mhead[s] = orc_mnode_new(o,
				    o->mnodes[m].pat, o->mhead[s]);
		}
	}

	/* convert_NFA_to_DFA, acsmx.c:444-486 */
	qh = qt = 0;
	for (i = 0; i < ORC_ALPHA; i++) {
		s = o->next[i];
		if (s)
			queue[qt++] = s;
	}
	while (qh < qt) {
		r = queue[qh++];
		for (i = 0; i < ORC_ALPHA; i++) {
			s = o->next[(size_t)r * ORC_ALPHA + i];
			if (s != ORC_FAIL)
				queue[qt++] = s;
			else
				o->next[(size_t)r * ORC_ALPHA + i] =
				    o->next[(size_t)o->fail[r] * ORC_ALPHA + i];
		}
	}
	free(queue);
}

/* acsmx.c:600-671.  Cells the reference leaves uninitialised (plane 1 of
 * non-final targets) are zero here. */
void
orc_gen_table(orc_t *o)
{
	int i, j, st;
	o->num_states += 1;
	o->table_states = o->num_states;
	o->table = (int32_t *)calloc((size_t)o->num_states * 2 * ORC_ALPHA,
	    sizeof(int32_t));
	if (!o->table) {
		fprintf(stderr, "oracle: out of memory (table)\n");
		exit(1);
	}
	for (i = 0; i < o->num_states; i++) {
		for (j = 0; j < ORC_ALPHA; j++) {
			st = o->next[(size_t)i * ORC_ALPHA + j];
			if (o->mhead[st] != -1) {
				o->table[(size_t)i * 512 + j] = -st;
				o->table[(size_t)i * 512 + 256 + j] =
				    o->mnodes[o->mhead[st]].pat;
			} else {
				o->table[(size_t)i * 512 + j] = st;
			}
		}
	}
}

int orc_num_states(const orc_t *o) { return o->num_states; }
int orc_num_patterns(const orc_t *o) { return o->npats; }
int orc_max_pattern_len(const orc_t *o) { return o->max_pattern_len; }
const int32_t *orc_table(const orc_t *o) { return o->table; }
const int32_t *orc_next_rows(const orc_t *o) { return o->next; }
const int32_t *orc_fail(const orc_t *o) { return o->fail; }

/* head-of-list pattern index of a state, or -1 */
int
orc_head_index(const orc_t *o, int state)
{
	return o->mhead[state] == -1 ? -1 : o->mnodes[o->mhead[state]].pat;
}

/* full match list of a state in list order; returns its length */
int
orc_match_list(const orc_t *o, int state, int *out, int cap)
{
	int m, n = 0;
	for (m = o->mhead[state]; m != -1; m = o->mnodes[m].next) {
		if (n < cap)
			out[n] = o->mnodes[m].pat;
		n++;
	}
	return n;
}

int
orc_pattern_info(const orc_t *o, int index, int *iid, int *n,
    unsigned char *bytes, int cap)
{
	if (index < 0 || index >= o->npats)
		return -1;
	*iid = o->pats[index].iid;
	*n = o->pats[index].n;
	if (bytes)
		memcpy(bytes, o->pats[index].bytes,
		    (size_t)(o->pats[index].n < cap ? o->pats[index].n : cap));
	return 0;
}

/* acsmx.c:677-735: next_chain[i] = index of the pattern linked after
 * pattern i (patterns that end in a common final state), or -1. */
void
orc_patterns_chain(const orc_t *o, int *next_chain)
{
	int i, m, q, hops;
	for (i = 0; i < o->npats; i++)
		next_chain[i] = -1;
	for (i = 0; i < o->max_states; i++) {
		m = o->mhead[i];
		if (m == -1 || o->mnodes[m].next == -1)
			continue;
		q = o->mnodes[m].pat;
		/* the reference walks "while (q->next)" unbounded and never
		 * returns once two states have linked a cycle (it hangs on
		 * its own apps/patterns.txt); bound the walk instead */
		for (hops = 0; next_chain[q] != -1 && hops < o->npats; hops++)
			q = next_chain[q];
		while (m != -1 && o->mnodes[m].next != -1) {
			int nx = o->mnodes[o->mnodes[m].next].pat;
			if (nx == q)
				break;
			next_chain[q] = nx;
			m = o->mnodes[m].next;
			q = next_chain[q];
		}
	}
}

/* FNV-1a over the *defined* cells of the reference-format table: plane 0
 * always, plane 1 only where plane 0 is negative. */
uint64_t
orc_table_digest(const int32_t *table, int nstates)
{
	uint64_t h = 1469598103934665603ULL;
	int i, j, k;
	for (i = 0; i < nstates; i++) {
		for (j = 0; j < ORC_ALPHA; j++) {
			int32_t v0 = table[(size_t)i * 512 + j];
			int32_t v1 = v0 < 0 ? table[(size_t)i * 512 + 256 + j]
					    : 0;
			uint32_t w[2] = { (uint32_t)v0, (uint32_t)v1 };
			for (k = 0; k < 8; k++) {
				h ^= (w[k >> 2] >> ((k & 3) * 8)) & 0xff;
				h *= 1099511628211ULL;
			}
		}
	}
	return h;
}

/* FNV-1a over a record stream (pos u32, pat i32), little endian. */
uint64_t
orc_records_digest(const uint32_t *pos, const int32_t *pat, size_t n)
{
	uint64_t h = 1469598103934665603ULL;
	size_t i;
	int k;
	for (i = 0; i < n; i++) {
		uint32_t w[2] = { pos[i], (uint32_t)pat[i] };
		for (k = 0; k < 8; k++) {
			h ^= (w[k >> 2] >> ((k & 3) * 8)) & 0xff;
			h *= 1099511628211ULL;
		}
	}
	return h;
}

/* -------------------------------------------------------- pattern loader */

static int
orc_half_hex(unsigned char c)
{
	if (isdigit(c))
		return c - '0';
	c = (unsigned char)tolower(c);
	if (c >= 'a' && c <= 'f')
		return c + 10 - 'a';
	return -1; /* reference: undefined (utils.c:18-26); we reject */
}

/*
 * ocl_worker.c:74-145.  hex: -x; max_len: -m (or -1).
 * Returns number of patterns added, or -1 on open/parse failure.
 * Deviations (documented): categorical detection implements
 * the intent ("[+-]?digits" then blank); non-hex digits are an error.
 */
int
orc_load_patterns(orc_t *o, const char *path, int hex, int max_len)
{
	FILE *fp = fopen(path, "r");
	char line[ORC_MAX_PAT_LINE];
	int i = 0, categ = 0;

	if (!fp)
		return -1;
	while (fgets(line, sizeof(line), fp)) {
		size_t len = strlen(line);
		char *pattern;
		size_t plen;
		long pat_id;

		if (len && line[len - 1] == '\n')
			line[--len] = '\0';

		if (i == 0) {
			size_t j = 0, k;
			categ = 0;
			while (j < len && line[j] != ' ' && line[j] != '\t')
				j++;
			if (j < len && j > 0) {
				k = (line[0] == '+' || line[0] == '-') ? 1 : 0;
				categ = (k < j);
				for (; k < j; k++)
					if (!isdigit((unsigned char)line[k]))
						categ = 0;
			}
		}
		if (categ) {
			char *end;
			errno = 0;
			pat_id = strtol(line, &end, 10);
			if (errno != 0) {
				fclose(fp);
				return -1;
			}
			while (isspace((unsigned char)*end))
				end++;
			pattern = end;
		} else {
			pattern = line;
			pat_id = i;
		}
		plen = strlen(pattern);
		if (plen >= 1 && pattern[0] == '"' && pattern[plen - 1] == '"') {
			if (plen >= 2) {
				pattern[plen - 1] = '\0';
				pattern++;
				plen -= 2;
			} else { /* a lone '"': reference yields length -1 */
				pattern[0] = '\0';
				plen = 0;
			}
		}
		if (hex) {
			unsigned char buf[ORC_MAX_PAT_LINE / 2 + 1];
			size_t k;
			if (max_len != -1 && (size_t)max_len * 2 < plen) {
				pattern[(size_t)max_len * 2] = '\0';
				plen = (size_t)max_len * 2;
			}
			if (plen % 2 != 0) {
				fclose(fp);
				return -1; /* utils.c:39-42 exits */
			}
			for (k = 0; k < plen; k += 2) {
				int hi = orc_half_hex((unsigned char)pattern[k]);
				int lo = orc_half_hex(
				    (unsigned char)pattern[k + 1]);
				if (hi < 0 || lo < 0) {
					fclose(fp);
					return -1;
				}
				buf[k / 2] = (unsigned char)(hi * 16 + lo);
			}
			orc_add_pattern(o, buf, (int)(plen / 2), (int)pat_id);
		} else {
			if (max_len != -1 && (size_t)max_len < plen) {
				pattern[max_len] = '\0';
				plen = (size_t)max_len;
			}
			orc_add_pattern(o, (unsigned char *)pattern, (int)plen,
			    (int)pat_id);
		}
		i++;
	}
	fclose(fp);
	return i;
}

/* ------------------------------------------------------------------ scan */

/*
 * Canonical serial semantics: one record (end offset,
 * head-of-list pattern index) per text position whose transition is
 * flagged final.  Returns the total number of matches; at most 'cap'
 * records are stored.  *final_state receives the state after the last byte.
 */
size_t
orc_scan_serial(const int32_t *table, const unsigned char *text, size_t n,
    long init_state, uint32_t *out_pos, int32_t *out_pat, size_t cap,
    long *final_state)
{
	size_t k, m = 0;
	long state = init_state, prev;
	for (k = 0; k < n; k++) {
		prev = state;
		state = table[(size_t)prev * 512 + text[k]];
		if (state < 0) {
			if (m < cap) {
				out_pos[m] = (uint32_t)k;
				out_pat[m] = table[(size_t)prev * 512 + 256 +
				    text[k]];
			}
			m++;
			state = -state;
		}
	}
	if (final_state)
		*final_state = state;
	return m;
}

/*
 * All-patterns variant: the same walk, but every pattern
 * of the entered state's match list is reported, in list order -- the list
 * acsmx.c builds at :299-312 / :417-429 and whose head the table carries
 * (:650).  The reference has no scanner that does this; its patterns table
 * links the entries for it (acsmx.c:707-721).
 */
size_t
orc_scan_serial_all(const orc_t *o, const unsigned char *text, size_t n,
    long init_state, uint32_t *out_pos, int32_t *out_pat, size_t cap,
    long *final_state)
{
	const int32_t *table = o->table;
	size_t k, m = 0;
	long state = init_state;
	int list[4096], len, j;
	for (k = 0; k < n; k++) {
		state = table[(size_t)state * 512 + text[k]];
		if (state < 0) {
			state = -state;
			len = orc_match_list(o, (int)state, list, 4096);
			for (j = 0; j < len && j < 4096; j++) {
				if (m < cap) {
					out_pos[m] = (uint32_t)k;
					out_pat[m] = list[j];
				}
				m++;
			}
		}
	}
	if (final_state)
		*final_state = state;
	return m;
}

/* count-only variant used for timing (no stores besides the counter) */
size_t
orc_scan_count(const int32_t *table, const unsigned char *text, size_t n,
    long init_state, long *final_state)
{
	size_t k, m = 0;
	long state = init_state;
	for (k = 0; k < n; k++) {
		state = table[(size_t)state * 512 + text[k]];
		if (state < 0) {
			m++;
			state = -state;
		}
	}
	if (final_state)
		*final_state = state;
	return m;
}

/* all-cores walk: contiguous shards with an (L-1)-byte halo, each walked
 * serially from state 0 (shard 0 from init_state); hits inside the halo are
 * dropped.  Records are concatenated in shard order => position order. */
typedef struct {
	const int32_t *table;
	const unsigned char *text;
	size_t begin, end, halo_begin;
	long init_state;
	uint32_t *pos;
	int32_t *pat;
	size_t cap, count;
	long final_state;
} orc_shard_t;

static void *
orc_shard_run(void *arg)
{
	orc_shard_t *s = (orc_shard_t *)arg;
	size_t k, m = 0;
	long state = s->init_state, prev;
	for (k = s->halo_begin; k < s->end; k++) {
		prev = state;
		state = s->table[(size_t)prev * 512 + s->text[k]];
		if (state < 0) {
			if (k >= s->begin) {
				if (m < s->cap) {
					s->pos[m] = (uint32_t)k;
					s->pat[m] = s->table[(size_t)prev * 512 +
					    256 + s->text[k]];
				}
				m++;
			}
			state = -state;
		}
	}
	s->count = m;
	s->final_state = state;
	return NULL;
}

size_t
orc_scan_threads(const int32_t *table, const unsigned char *text, size_t n,
    long init_state, int max_pat_len, int nthreads, uint32_t *out_pos,
    int32_t *out_pat, size_t cap, long *final_state)
{
	orc_shard_t *sh;
	pthread_t *th;
	size_t total = 0, halo = max_pat_len > 0 ? (size_t)max_pat_len - 1 : 0;
	int t;

	if (nthreads < 1)
		nthreads = 1;
	sh = (orc_shard_t *)calloc((size_t)nthreads, sizeof(*sh));
	th = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
	for (t = 0; t < nthreads; t++) {
		sh[t].table = table;
		sh[t].text = text;
		sh[t].begin = n * (size_t)t / (size_t)nthreads;
		sh[t].end = n * (size_t)(t + 1) / (size_t)nthreads;
		if (t == 0 || sh[t].begin < halo) {
			/* not enough room for a full halo: start from the
			 * very beginning with the caller's state */
			sh[t].halo_begin = 0;
			sh[t].init_state = init_state;
		} else {
			sh[t].halo_begin = sh[t].begin - halo;
			sh[t].init_state = 0;
		}
		sh[t].cap = cap; /* private worst-case buffers below */
		sh[t].pos = (uint32_t *)malloc(
		    (sh[t].end - sh[t].begin + 1) * sizeof(uint32_t));
		sh[t].pat = (int32_t *)malloc(
		    (sh[t].end - sh[t].begin + 1) * sizeof(int32_t));
		sh[t].cap = sh[t].end - sh[t].begin + 1;
		pthread_create(&th[t], NULL, orc_shard_run, &sh[t]);
	}
	for (t = 0; t < nthreads; t++) {
		size_t i;
		pthread_join(th[t], NULL);
		for (i = 0; i < sh[t].count; i++) {
			if (total < cap) {
				out_pos[total] = sh[t].pos[i];
				out_pat[total] = sh[t].pat[i];
			}
			total++;
		}
		free(sh[t].pos);
		free(sh[t].pat);
	}
	if (final_state)
		*final_state = sh[nthreads - 1].final_state;
	free(sh);
	free(th);
	return total;
}

/*
 * Reference kernel semantics (ahomatch.cl:1-165), emulated work-item by
 * work-item.  Only for documenting F2/Q8 (duplicates/losses at chunk
 * borders); the product targets orc_scan_serial.  results/results2 are the
 * bucket planes [max_results][chunks] + 1.
 */
void
orc_scan_refkernel(const int32_t *trans, const unsigned char *data,
    const int *indices, const int *sizes, int *results, int *results2,
    unsigned chunks, unsigned long data_size, long last_state,
    int max_pat_size, int max_results)
{
	unsigned id;
	for (id = 0; id < chunks; id++) {
		int index = indices[id], size = sizes[id];
		int matches = 0, i = 0, j;
		long state = (id == 0) ? last_state : 0, prev;
		size = (size + 15) / 16;
		for (i = 0; i < size; i++) {
			for (j = 0; j < 16; j++) {
				unsigned char c = data[(size_t)index / 16 * 16 +
				    (size_t)i * 16 + j];
				prev = state;
				state = trans[512 * (size_t)state + c];
				if (state < 0) {
					matches++;
					state = -state;
					if (matches < max_results) {
						results[matches * chunks + id] =
						    trans[512 * (size_t)prev +
						    c + 256];
						results2[matches * chunks + id] =
						    index + i * 16 + j;
					}
				}
			}
		}
		if (id == chunks - 1) {
			results[chunks * max_results] = (int)state;
			goto end;
		}
		if (state == 0)
			goto end;
		size += (max_pat_size + 15) / 16;
		for (; i < size; i++) {
			if ((unsigned long)i * 16 + index + 16 > data_size)
				goto end;
			for (j = 0; j < 16; j++) {
				unsigned char c = data[(size_t)index / 16 * 16 +
				    (size_t)i * 16 + j];
				prev = state;
				state = trans[512 * (size_t)state + c];
				if (state == 0)
					goto end;
				if (state < 0) {
					matches++;
					state = -state;
					if (matches < max_results) {
						results[matches * chunks + id] =
						    trans[512 * (size_t)prev +
						    c + 256];
						results2[matches * chunks + id] =
						    index + i * 16 + j;
					}
					goto end;
				}
			}
		}
end:
		results[id] = matches;
		results2[id] = matches;
	}
}

/* ------------------------------------------------- result post-processing */

/* exclusive prefix sum, int32 (what ocl_prefix_sum computes on the per-chunk
 * counts; the reference's float typing is quirk Q13, not reproduced) */
void
orc_exclusive_scan(const int32_t *in, int32_t *out, size_t n)
{
	int32_t acc = 0;
	size_t i;
	for (i = 0; i < n; i++) {
		out[i] = acc;
		acc += in[i];
	}
}

/* compactarray.cl:40-68 run for every gid < len */
void
orc_compact_array(int32_t *dst, const int32_t *src, const int32_t *prefix,
    int len, int max_results)
{
	int gid, i;
	dst[0] = prefix[len - 1] + src[len - 1];
	dst[dst[0] + 1] = src[(size_t)max_results * len];
	for (gid = 0; gid < len; gid++) {
		int off = prefix[gid], m = src[gid];
		for (i = 0; i < m && i < max_results - 1; i++)
			dst[off + 1 + i] = src[(size_t)len * (i + 1) + gid];
	}
}

/* key/value sort on unsigned keys; dir != 0 ascending (BitonicSort.cl:19-45
 * comparator: swap when (keyA > keyB) == dir).  The network is restated
 * stage by stage so that the value order among equal keys matches too:
 *   - sub-sorts up to LOCAL_SIZE_LIMIT=512 use dir = (i & size/2) != 0
 *     irrespective of sortDir (BitonicSort.cl:74-85, :135-146; the 512 stage
 *     of bitonicSortLocal1 uses group&1 == (i & 256) != 0, :150-161),
 *   - larger merges use sortDir ^ ((i & size/2) != 0) (:186, :229),
 *   - the last stage (size == len) uses sortDir (:88-99; for len > 512 the
 *     xor term is 0 because i < len/2).
 * Host-side checks as ocl_bitonic_sort.c:149-165. */
int
orc_bitonic_sort(uint32_t *key, uint32_t *val, unsigned batch, unsigned len,
    unsigned dir)
{
	unsigned b, size, stride, i;
	if (len < 2)
		return 0;
	if (len & (len - 1))
		return -1;
	if (len <= 512 && ((size_t)batch * len) % 512 != 0)
		return -1;
	dir = (dir != 0);
	for (b = 0; b < batch; b++) {
		uint32_t *k = key + (size_t)b * len, *v = val + (size_t)b * len;
		for (size = 2; size <= len; size <<= 1) {
			for (stride = size / 2; stride > 0; stride >>= 1) {
				for (i = 0; i < len / 2; i++) {
					unsigned pos = 2 * i - (i & (stride - 1));
					unsigned d = dir;
					if (size < len) {
						d = ((i & (size / 2)) != 0);
						if (size > 512)
							d ^= dir;
					}
					if ((k[pos] > k[pos + stride]) == d) {
						uint32_t t = k[pos];
						k[pos] = k[pos + stride];
						k[pos + stride] = t;
						t = v[pos];
						v[pos] = v[pos + stride];
						v[pos + stride] = t;
					}
				}
			}
		}
	}
	return 0;
}

/* databuf.c:747-782: walk bucket planes, call cb(file, pat, chunk, off+1) */
typedef int (*orc_match_cb)(int file_idx, int patrn_idx, int chunk_idx,
    int offset, void *uarg);

int
orc_process_buckets(const int *res, const int *res2, const int *file_ids,
    int chunks, int max_results, orc_match_cb cb, void *uarg)
{
	int i, j, matches = 0;
	for (i = 0; i < chunks; i++) {
		matches += res[i];
		for (j = 0; j < res[i] && j < max_results - 1; j++) {
			if (cb)
				cb(file_ids[i], res[(size_t)(j + 1) * chunks + i],
				    i, res2[(size_t)(j + 1) * chunks + i] + 1,
				    uarg);
		}
	}
	return matches;
}

/* Fill the reference bucket planes from a position-ordered record list under
 * the canonical serial semantics: chunk i owns text positions
 * [indices[i], indices[i] + sizes[i]).  Cell 0 of each column = number of
 * records in the chunk (all of them, like the kernel's 'matches'), rows
 * 1..max_results-1 = the first records, trailer cell = last state. */
void
orc_bucketize(const uint32_t *pos, const int32_t *pat, size_t m,
    const int *indices, const int *sizes, int chunks, int max_results,
    long last_state, int *results, int *results2)
{
	size_t r = 0;
	int i;
	for (i = 0; i < chunks; i++) {
		int cnt = 0;
		uint32_t lo = (uint32_t)indices[i];
		uint32_t hi = lo + (uint32_t)sizes[i];
		while (r < m && pos[r] < lo)
			r++; /* positions in inter-chunk padding */
		while (r < m && pos[r] < hi) {
			cnt++;
			if (cnt < max_results) {
				results[(size_t)cnt * chunks + i] = pat[r];
				results2[(size_t)cnt * chunks + i] = (int)pos[r];
			}
			r++;
		}
		results[i] = cnt;
		results2[i] = cnt;
	}
	results[(size_t)chunks * max_results] = (int)last_state;
}

/* ---------------------------------------------------------------- timing */

double
orc_now(void)
{
	struct timespec tp;
	clock_gettime(CLOCK_MONOTONIC, &tp); /* utils.c:60-68 gettime() */
	return (double)tp.tv_sec + (double)tp.tv_nsec * 1e-9;
}
