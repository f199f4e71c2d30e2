// Host-side Aho-Corasick automaton: pattern list -> trie -> fail links ->
// match lists -> device numbering -> dense DFA rows.
//
// Semantics follow the reference's acsmx.c exactly where they are observable
// (state creation order, match-list order, which pattern a final state
// reports); the data structures do not: the trie is sparse (CSR children +
// hash lookup), fail links are computed on the trie, and dense rows are
// generated once, directly in device numbering.
//
//   acsm_add_pattern   acsmx.c:514-546   -> acm_automaton_add
//   add_pattern_states acsmx.c:318-349   -> insert_patterns()
//   build_NFA          acsmx.c:355-438   -> link_and_collect()
//   convert_NFA_to_DFA acsmx.c:444-486   -> dense_rows()
//   acsm_gen_state_table :640-658        -> acm_automaton_export_reference_table
//   acsm_get_patterns_table :677-735     -> chain_patterns()
//   pattern file parser ocl_worker.c:74-145, utils.c:18-54 -> acm_automaton_load_file
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <unordered_map>

#include "acm_internal.h"

namespace acm {

static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}

void clear_error() { g_last_error.clear(); }

}  // namespace acm

extern "C" const char *acm_last_error(void) { return acm::g_last_error.c_str(); }

extern "C" const char *acm_strerror(int code)
{
	switch (code) {
	case ACM_OK: return "ok";
	case ACM_ERR_ARG: return "invalid argument";
	case ACM_ERR_NOMEM: return "out of memory";
	case ACM_ERR_HIP: return "HIP runtime error";
	case ACM_ERR_NODEV: return "no usable device";
	case ACM_ERR_LIMIT: return "design limit exceeded";
	case ACM_ERR_IO: return "cannot open file";
	case ACM_ERR_PARSE: return "malformed pattern";
	case ACM_ERR_CAPACITY: return "result planes too small";
	default: return "unknown error";
	}
}

extern "C" const char *acm_version(void) { return "acmatch 0.1 (gfx950, HIP)"; }

// ---------------------------------------------------------------------------

extern "C" acm_automaton *acm_automaton_new(void)
{
	return new (std::nothrow) acm_automaton();
}

extern "C" void acm_automaton_free(acm_automaton *a) { delete a; }

extern "C" int acm_automaton_add(acm_automaton *a, const unsigned char *bytes, int n, int iid)
{
	if (!a || n < 0 || (n > 0 && !bytes))
		return acm::fail(ACM_ERR_ARG, "acm_automaton_add: bad arguments");
	if (a->compiled)
		return acm::fail(ACM_ERR_ARG, "acm_automaton_add: automaton already compiled");
	acm_automaton::Pattern p;
	p.bytes.assign(bytes, bytes + n);
	p.iid = iid;
	a->patterns.push_back(std::move(p));
	if (n > a->max_pattern_len)
		a->max_pattern_len = n;
	return ACM_OK;
}

// ---- pattern file ------------------------------------------------------------

static int hex_nibble(unsigned char c)
{
	if (c >= '0' && c <= '9') return c - '0';
	if (c >= 'a' && c <= 'f') return c - 'a' + 10;
	if (c >= 'A' && c <= 'F') return c - 'A' + 10;
	return -1;
}

// "[+-]?digits" followed by a blank => the file is "ID pattern" per line.
// The reference decides this on line 0 only (ocl_worker.c:79-102); its scan
// loop is broken (tests i instead of j), we implement what it means to do.
static bool looks_categorical(const std::string &line)
{
	size_t sp = line.find_first_of(" \t");
	if (sp == std::string::npos || sp == 0)
		return false;
	size_t k = (line[0] == '+' || line[0] == '-') ? 1 : 0;
	if (k >= sp)
		return false;
	for (; k < sp; k++)
		if (!isdigit((unsigned char)line[k]))
			return false;
	return true;
}

extern "C" int acm_automaton_load_file(acm_automaton *a, const char *path, int hex, int max_len)
{
	if (!a || !path)
		return acm::fail(ACM_ERR_ARG, "acm_automaton_load_file: bad arguments");
	FILE *fp = fopen(path, "r");
	if (!fp)
		return acm::fail(ACM_ERR_IO, "cannot open pattern file '%s': %s", path, strerror(errno));

	// The reference reads with fgets into a 4096-byte buffer, so a longer
	// line arrives in pieces, each piece a pattern of its own (Q18).  Keep
	// that: it is observable through pattern indices.
	std::vector<char> buf(acm::kMaxPatternLine);
	bool categorical = false;
	int line_no = 0, rc = ACM_OK;
	while (fgets(buf.data(), (int)buf.size(), fp)) {
		std::string line(buf.data());
		if (!line.empty() && line.back() == '\n')
			line.pop_back();
		if (line_no == 0)
			categorical = looks_categorical(line);

		long iid = line_no;
		std::string pat = line;
		if (categorical) {
			char *end = nullptr;
			errno = 0;
			iid = strtol(line.c_str(), &end, 10);
			if (errno != 0) {
				rc = acm::fail(ACM_ERR_PARSE, "%s:%d: bad pattern id", path, line_no + 1);
				break;
			}
			while (*end && isspace((unsigned char)*end))
				end++;
			pat.assign(end);
		}
		if (pat.size() >= 2 && pat.front() == '"' && pat.back() == '"')
			pat = pat.substr(1, pat.size() - 2);
		else if (pat.size() == 1 && pat[0] == '"')
			pat.clear();

		if (hex) {
			if (max_len != -1 && pat.size() > (size_t)max_len * 2)
				pat.resize((size_t)max_len * 2);
			if (pat.size() % 2) {  // utils.c:39-42 prints and exits
				rc = acm::fail(ACM_ERR_PARSE, "%s:%d: odd number of hex digits", path,
				    line_no + 1);
				break;
			}
			std::vector<unsigned char> bytes(pat.size() / 2);
			bool ok = true;
			for (size_t k = 0; k < bytes.size(); k++) {
				int hi = hex_nibble((unsigned char)pat[2 * k]);
				int lo = hex_nibble((unsigned char)pat[2 * k + 1]);
				if (hi < 0 || lo < 0) {
					ok = false;
					break;
				}
				bytes[k] = (unsigned char)(hi * 16 + lo);
			}
			if (!ok) {
				rc = acm::fail(ACM_ERR_PARSE, "%s:%d: not a hex digit", path, line_no + 1);
				break;
			}
			rc = acm_automaton_add(a, bytes.data(), (int)bytes.size(), (int)iid);
		} else {
			if (max_len != -1 && pat.size() > (size_t)max_len)
				pat.resize((size_t)max_len);
			rc = acm_automaton_add(a, (const unsigned char *)pat.data(), (int)pat.size(), (int)iid);
		}
		if (rc != ACM_OK)
			break;
		line_no++;
	}
	fclose(fp);
	return rc == ACM_OK ? line_no : rc;
}

// ---- compile -------------------------------------------------------------------

namespace {

struct Builder {
	acm_automaton &a;
	std::unordered_map<uint64_t, uint32_t> edge;  // (state << 8 | byte) -> child
	std::vector<std::vector<int32_t>> own;        // per state, patterns in the order they were attached

	explicit Builder(acm_automaton &aut) : a(aut) {}

	uint32_t child(uint32_t s, uint8_t c) const
	{
		auto it = edge.find(((uint64_t)s << 8) | c);
		return it == edge.end() ? UINT32_MAX : it->second;
	}

	// Newest pattern first: acsm_add_pattern prepends (acsmx.c:536-538) and
	// acsm_compile walks from the head (:579-580), so the last line of the
	// pattern file creates states 1..n.
	void insert_patterns()
	{
		size_t total = 1;
		for (auto &p : a.patterns)
			total += p.bytes.size();
		edge.reserve(total * 2);
		a.parent.assign(1, 0);
		a.in_byte.assign(1, 0);
		a.depth.assign(1, 0);
		own.assign(1, {});
		for (int i = (int)a.patterns.size() - 1; i >= 0; i--) {
			const auto &bytes = a.patterns[i].bytes;
			uint32_t s = 0;
			size_t k = 0;
			for (; k < bytes.size(); k++) {
				uint32_t t = child(s, bytes[k]);
				if (t == UINT32_MAX)
					break;
				s = t;
			}
			for (; k < bytes.size(); k++) {
				uint32_t t = (uint32_t)a.parent.size();
				edge.emplace(((uint64_t)s << 8) | bytes[k], t);
				a.parent.push_back(s);
				a.in_byte.push_back(bytes[k]);
				a.depth.push_back((uint16_t)(a.depth[s] + 1));
				own.emplace_back();
				s = t;
			}
			own[s].push_back(i);
		}
		a.num_states = (uint32_t)a.parent.size();
	}

	void index_children()
	{
		uint32_t n = a.num_states;
		a.child_begin.assign(n + 1, 0);
		for (uint32_t s = 1; s < n; s++)
			a.child_begin[a.parent[s] + 1]++;
		for (uint32_t s = 0; s < n; s++)
			a.child_begin[s + 1] += a.child_begin[s];
		a.child_list.resize(n ? n - 1 : 0);
		std::vector<uint32_t> cursor(a.child_begin.begin(), a.child_begin.end() - 1);
		for (uint32_t s = 1; s < n; s++)
			a.child_list[cursor[a.parent[s]]++] = { a.in_byte[s], s };
		for (uint32_t s = 0; s < n; s++)
			std::sort(a.child_list.begin() + a.child_begin[s],
			    a.child_list.begin() + a.child_begin[s + 1],
			    [](const acm_automaton::Edge &x, const acm_automaton::Edge &y) {
				    return x.byte < y.byte;
			    });
	}

	// BFS: fail links on the trie, and the match list of every state:
	//   list(s) = reverse(list(fail(s))) ++ own(s), own(s) newest first.
	// That is what build_NFA's "copy every node of the fail state's list,
	// each pushed at the front" produces (acsmx.c:417-429) on top of the
	// push-front of add_match_list_entry (:304-309).
	void link_and_collect()
	{
		uint32_t n = a.num_states;
		a.fail.assign(n, 0);
		a.list_begin.assign(n, -1);
		a.list_len.assign(n, 0);
		a.list_pool.clear();
		a.bfs_order.clear();
		a.bfs_order.reserve(n);
		a.bfs_order.push_back(0);
		set_list(0, 0, false);
		for (size_t qh = 0; qh < a.bfs_order.size(); qh++) {
			uint32_t s = a.bfs_order[qh];
			for (uint32_t e = a.child_begin[s]; e < a.child_begin[s + 1]; e++) {
				uint32_t t = a.child_list[e].to;
				uint8_t c = a.child_list[e].byte;
				uint32_t f = 0;
				if (s != 0) {
					uint32_t g = a.fail[s];
					for (;;) {
						uint32_t u = child(g, c);
						if (u != UINT32_MAX) {
							f = u;
							break;
						}
						if (g == 0)
							break;
						g = a.fail[g];
					}
				}
				a.fail[t] = f;
				// depth-1 states get fail = root WITHOUT inheriting the
				// root's list (acsmx.c:376-382 vs :417-429); it only
				// matters when an empty pattern made the root "final"
				set_list(t, f, /*inherit=*/s != 0);
				a.bfs_order.push_back(t);
			}
		}
	}

	void set_list(uint32_t s, uint32_t f, bool inherit)
	{
		int inherited = inherit ? a.list_len[f] : 0;
		int total = inherited + (int)own[s].size();
		if (total == 0)
			return;
		a.list_begin[s] = (int32_t)a.list_pool.size();
		a.list_len[s] = total;
		for (int k = inherited - 1; k >= 0; k--) {
			int32_t v = a.list_pool[a.list_begin[f] + k];
			a.list_pool.push_back(v);
		}
		for (int k = (int)own[s].size() - 1; k >= 0; k--)
			a.list_pool.push_back(own[s][k]);
	}

	// dev ids: hot rows first (the first non-final states in BFS order), then
	// the other non-final states and the final states, both in ref order.
	// The root is never final: a transition into state 0 cannot be flagged
	// because the reference stores finals as -state (acsmx.c:645).
	void byte_classes()
	{
		bool used[256] = { false };
		for (const auto &p : a.patterns)
			for (uint8_t b : p.bytes)
				used[b] = true;
		uint32_t nused = 0;
		for (int b = 0; b < 256; b++)
			nused += used[b] ? 1 : 0;
		const uint32_t classes = nused + (nused < 256 ? 1 : 0);
		if (classes > 128) {   // nothing to gain: the identity
			for (int b = 0; b < 256; b++)
				a.byte_class[b] = a.class_byte[b] = (uint8_t)b;
			a.num_classes = 256;
			a.log_stride = 8;
			return;
		}
		uint32_t next = 1, spare = 0;
		for (int b = 0; b < 256; b++)
			if (!used[b])
				spare = (uint32_t)b;   // (one exists: fewer than 256 bytes are used)
		memset(a.class_byte, (int)spare, sizeof(a.class_byte));
		for (int b = 0; b < 256; b++) {
			a.byte_class[b] = used[b] ? (uint8_t)next : 0;
			if (used[b])
				a.class_byte[next++] = (uint8_t)b;
		}
		a.num_classes = classes;
		a.log_stride = 1;
		while ((1u << a.log_stride) < classes)
			a.log_stride++;
	}

	void number_for_device()
	{
		const uint32_t n = a.num_states;
		// rows the LDS budget of the walk kernel holds (a hot cell is 16 bits: ids below the sentinel)
		const uint32_t hot_max = std::min<uint32_t>(acm::kHotBytes / (2u << a.log_stride), 0x8000u);
		a.ref2dev.assign(n, UINT32_MAX);
		a.dev2ref.assign(n, 0);
		uint32_t next = 0;
		a.hot_depth1 = 0;
		for (uint32_t s : a.bfs_order) {
			if (next >= hot_max)
				break;
			if (a.is_final_ref(s))
				continue;
			if (a.depth[s] <= 1)
				a.hot_depth1 = next + 1;
			a.ref2dev[s] = next;
			a.dev2ref[next++] = s;
		}
		a.hot_count = next;
		for (uint32_t s = 0; s < n; s++)
			if (a.ref2dev[s] == UINT32_MAX && !a.is_final_ref(s)) {
				a.ref2dev[s] = next;
				a.dev2ref[next++] = s;
			}
		a.first_final = next;
		for (uint32_t s = 0; s < n; s++)
			if (a.ref2dev[s] == UINT32_MAX) {
				a.ref2dev[s] = next;
				a.dev2ref[next++] = s;
			}
	}

	// dev_run[d]: length of the unary, non-final path d -> d+1 -> d+2 ...
	// Computed right to left over dev ids.
	void unary_runs()
	{
		const uint32_t n = a.num_states;
		a.dev_run.assign(n, 0);
		for (uint32_t d = n; d-- > 0;) {
			if (d + 1 >= a.first_final || d == 0)
				continue;   // the state entered must not be final; the root has no run
			const uint32_t r = a.dev2ref[d], r1 = a.dev2ref[d + 1];
			const bool only_child = a.child_begin[r + 1] - a.child_begin[r] == 1 &&
			    a.child_list[a.child_begin[r]].to == r1;
			if (!only_child)
				continue;
			const uint32_t nx = a.dev_run[d + 1];
			a.dev_run[d] = (uint16_t)(nx >= 65534 ? 65535 : nx + 1);
		}
	}

	// acsm_get_patterns_table's linking pass (acsmx.c:707-721), state ids
	// ascending.  The reference's "walk to the end of q's chain" never ends
	// once two states have linked their patterns into a cycle -- it does so
	// on its own apps/patterns.txt -- so the walk is bounded by the number
	// of patterns and simply stops where it is when it does not terminate.
	void chain_patterns()
	{
		a.next_chained.assign(a.patterns.size(), -1);
		const size_t bound = a.patterns.size();
		for (uint32_t s = 0; s < a.num_states; s++) {
			if (a.list_len[s] < 2)
				continue;
			const int32_t *l = &a.list_pool[a.list_begin[s]];
			int32_t q = l[0];
			for (size_t hops = 0; a.next_chained[q] != -1 && hops < bound; hops++)
				q = a.next_chained[q];
			for (int k = 0; k + 1 < a.list_len[s]; k++) {
				if (l[k + 1] == q)
					break;
				a.next_chained[q] = l[k + 1];
				q = l[k + 1];
			}
		}
	}
};

}  // namespace

extern "C" int acm_automaton_compile(acm_automaton *a)
{
	if (!a)
		return acm::fail(ACM_ERR_ARG, "acm_automaton_compile: null automaton");
	if (a->compiled)
		return ACM_OK;
	if (a->max_pattern_len >= acm::kMaxPatternLine)
		return acm::fail(ACM_ERR_LIMIT, "pattern longer than %d bytes", acm::kMaxPatternLine - 1);
	size_t total = 1;
	for (auto &p : a->patterns)
		total += p.bytes.size();
	if (total > acm::kMaxStates)
		return acm::fail(ACM_ERR_LIMIT, "pattern set may need %zu states (limit %u)", total,
		    acm::kMaxStates);
	try {
		Builder b(*a);
		b.insert_patterns();
		b.index_children();
		b.link_and_collect();
		b.byte_classes();
		b.number_for_device();
		b.chain_patterns();
		b.unary_runs();
	} catch (const std::bad_alloc &) {
		return acm::fail(ACM_ERR_NOMEM, "out of memory while compiling the automaton");
	}
	a->compiled = true;
	return ACM_OK;
}

// Dense rows in BFS order: row(s) = row(fail(s)) with the trie children of s
// written over it; the root row is all "-> root" plus its children.  fail(s)
// is shallower than s, so its row exists by the time s is reached.  A cell
// carries everything the deep walks need to know about its target.
const std::vector<uint64_t> &acm_automaton::dense_rows() const
{
	if (!dense.empty() || num_states == 0)
		return dense;
	std::vector<uint64_t> cell(num_states);
	for (uint32_t r = 0; r < num_states; r++) {
		const uint32_t d = ref2dev[r];
		cell[r] = (uint64_t)d | ((uint64_t)depth[r] << 32) | ((uint64_t)dev_run[d] << 48);
	}
	dense.assign((size_t)num_states * 256, cell[0]);
	for (uint32_t s : bfs_order) {
		uint64_t *row = &dense[(size_t)ref2dev[s] * 256];
		if (s != 0)
			memcpy(row, &dense[(size_t)ref2dev[fail[s]] * 256], 256 * sizeof(uint64_t));
		for (uint32_t e = child_begin[s]; e < child_begin[s + 1]; e++)
			row[child_list[e].byte] = cell[child_list[e].to];
	}
	return dense;
}

// ---- accessors -------------------------------------------------------------------

extern "C" int acm_automaton_num_patterns(const acm_automaton *a)
{
	return a ? (int)a->patterns.size() : 0;
}

extern "C" int acm_automaton_max_pattern_len(const acm_automaton *a)
{
	return a ? a->max_pattern_len : 0;
}

extern "C" int acm_automaton_byte_classes(const acm_automaton *a, uint8_t *class_of)
{
	if (!a || !a->compiled)
		return 0;
	if (class_of)
		memcpy(class_of, a->byte_class, 256);
	return (int)a->num_classes;
}

extern "C" int acm_automaton_num_states(const acm_automaton *a)
{
	return (a && a->compiled) ? (int)a->num_states : 0;
}

extern "C" size_t acm_automaton_reference_table_bytes(const acm_automaton *a)
{
	return (a && a->compiled) ? (size_t)a->num_states * 2 * 256 * sizeof(int32_t) : 0;
}

extern "C" int acm_automaton_export_reference_table(const acm_automaton *a, int32_t *dst)
{
	if (!a || !a->compiled || !dst)
		return acm::fail(ACM_ERR_ARG, "export_reference_table: automaton not compiled");
	try {
		const std::vector<uint64_t> &rows = a->dense_rows();
		for (uint32_t s = 0; s < a->num_states; s++) {
			const uint64_t *row = &rows[(size_t)a->ref2dev[s] * 256];
			int32_t *out = dst + (size_t)s * 512;
			for (int c = 0; c < 256; c++) {
				const uint32_t td = (uint32_t)row[c];
				uint32_t t = a->dev2ref[td];
				if (td >= a->first_final) {
					out[c] = -(int32_t)t;
					out[256 + c] = a->head_of(t);
				} else {
					out[c] = (int32_t)t;
					out[256 + c] = 0;
				}
			}
		}
	} catch (const std::bad_alloc &) {
		return acm::fail(ACM_ERR_NOMEM, "out of memory building dense rows");
	}
	return ACM_OK;
}

extern "C" int acm_automaton_pattern(const acm_automaton *a, int index, int *iid, int *n,
    const unsigned char **bytes, int *next_chained)
{
	if (!a || index < 0 || index >= (int)a->patterns.size())
		return acm::fail(ACM_ERR_ARG, "acm_automaton_pattern: index out of range");
	if (iid) *iid = a->patterns[index].iid;
	if (n) *n = (int)a->patterns[index].bytes.size();
	if (bytes) *bytes = a->patterns[index].bytes.data();
	if (next_chained)
		*next_chained = a->compiled ? a->next_chained[index] : -1;
	return ACM_OK;
}

extern "C" int acm_automaton_state_matches(const acm_automaton *a, int ref_state, int32_t *out, int cap)
{
	if (!a || !a->compiled || ref_state < 0 || (uint32_t)ref_state >= a->num_states || (cap > 0 && !out))
		return -1;
	if (!a->is_final_ref((uint32_t)ref_state))
		return 0;
	const int32_t len = a->list_len[ref_state];
	for (int32_t i = 0; i < len && i < cap; i++)
		out[i] = a->list_pool[(size_t)a->list_begin[ref_state] + i];
	return len;
}

extern "C" int acm_automaton_state_output(const acm_automaton *a, int ref_state)
{
	if (!a || !a->compiled || ref_state < 0 || (uint32_t)ref_state >= a->num_states)
		return -1;
	return a->head_of((uint32_t)ref_state);
}
