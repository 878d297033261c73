// tables.cpp -- alphabets, codon table, BLOSUM62 and option defaults.
//
// Replaces nasw-tab.c:6-107 (tables), nasw-tab.c:59-83 (matrix), options.c:10-90 (defaults).
// The alphabet ORDER is part of the interface (scores index the 22x22 matrix by it, the reduced
// alphabet feeds the k-mer hash that the .mpi index was built with), so it is the reference's.
#include <cstring>
#include <cmath>
#include <mutex>
#include <chrono>
#include <vector>
#include <string>
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include "mpa_internal.h"

namespace mpa {


const char *const kAA = "ARNDCQEGHILKMFPSTWYV*X";

} // namespace mpa

// The tables are the reference's public globals as well (miniprot.h:149-151, nasw.h:55-58): a program written against the
// reference reads ns_tab_* directly, so they are the storage and the mpa:: accessors point at them.
extern "C" {
uint8_t ns_tab_nt4[256], ns_tab_aa20[256], ns_tab_aa13[256], ns_tab_codon[64], ns_tab_codon13[64];
// reduced 4-bit alphabet used for seeding (nasw-tab.c:12): similar residues share a code, '*'=14, 'X'=15
uint8_t ns_tab_a2r[22] = { 0, 2, 4, 4, 6, 5, 5, 8, 3, 10, 11, 2, 11, 12, 7, 1, 1, 13, 12, 10, 14, 15 };
static char kNtI2c[] = "ACGTN", kAaI2c[] = "ARNDCQEGHILKMFPSTWYV*X";
char *ns_tab_nt_i2c = kNtI2c, *ns_tab_aa_i2c = kAaI2c;
}

namespace mpa {
static uint8_t *const g_nt4 = ns_tab_nt4, *const g_aa20 = ns_tab_aa20, *const g_aa13 = ns_tab_aa13, *const g_codon = ns_tab_codon, *const g_codon13 = ns_tab_codon13;
static std::once_flag g_once;
static int g_code = 1;

// NCBI genetic codes as their "AAs" strings in TCAG order (first base slowest).
static const char *ncbi_aas(int code)
{
	switch (code) {
	case 1: case 11: return "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 2:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG";
	case 3:  return "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 4:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 5:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG";
	case 6:  return "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 9:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
	case 10: return "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 12: return "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 13: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG";
	case 14: return "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
	case 15: return "FFLLSSSSYY*QCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 16: return "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 21: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
	case 22: return "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 23: return "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 24: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG";
	case 25: return "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 26: return "FFLLSSSSYY**CC*WLLLAPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 27: case 28: return "FFLLSSSSYYQQCCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 29: return "FFLLSSSSYYYYCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 30: return "FFLLSSSSYYEECC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 31: return "FFLLSSSSYYEECCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 32: return "FFLLSSSSYY*WCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
	case 33: return "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG";
	default: return nullptr;
	}
}

static void fill_codon(const char *aas)
{
	static const int tcag[4] = { 3, 1, 0, 2 };           // position in "TCAG" -> nt4 code
	for (int a = 0; a < 4; ++a)
		for (int b = 0; b < 4; ++b)
			for (int c = 0; c < 4; ++c) {
				int idx = tcag[a] << 4 | tcag[b] << 2 | tcag[c];
				g_codon[idx] = g_aa20[(uint8_t)aas[a * 16 + b * 4 + c]];
				g_codon13[idx] = ns_tab_a2r[g_codon[idx]];
			}
}

static void init_tables()
{
	memset(g_nt4, 4, 256);
	const char *nts = "ACGTN";
	for (int i = 0; i < 5; ++i)
		g_nt4[i] = g_nt4[(uint8_t)nts[i]] = g_nt4[(uint8_t)(nts[i] | 0x20)] = (uint8_t)i;
	memset(g_aa20, 21, 256);
	memset(g_aa13, 15, 256);
	for (int i = 0; i < 22; ++i) {
		uint8_t up = (uint8_t)kAA[i], lo = (up >= 'A' && up <= 'Z') ? (uint8_t)(up | 0x20) : up;
		g_aa20[i] = g_aa20[up] = g_aa20[lo] = (uint8_t)i;
		g_aa13[i] = g_aa13[up] = g_aa13[lo] = ns_tab_a2r[i];
	}
	fill_codon(ncbi_aas(1));
}

static void ensure() { std::call_once(g_once, init_tables); }
// a program that reads ns_tab_* directly expects them filled without calling into the library first
__attribute__((constructor)) static void tables_at_load() { ensure(); }
const uint8_t *tab_nt4() { ensure(); return g_nt4; }
const uint8_t *tab_aa20() { ensure(); return g_aa20; }
const uint8_t *tab_aa13() { ensure(); return g_aa13; }
const uint8_t *tab_codon() { ensure(); return g_codon; }
const uint8_t *tab_codon13() { ensure(); return g_codon13; }

int set_trans_code(int code)
{
	ensure();
	const char *aas = ncbi_aas(code);
	if (aas == nullptr) { set_error("translation table " + std::to_string(code) + " is not available in this build"); return -2; }
	fill_codon(aas);
	g_code = code;
	return 0;
}

// BLOSUM62 over ARNDCQEGHILKMFPSTWYV*X (the NCBI matrix re-ordered to the alphabet above; '*' vs
// anything = -4, '*' vs '*' = 1, X column from the NCBI matrix).
} // namespace mpa
extern "C" int8_t ns_mat_blosum62[484] = {
	 4,-1,-2,-2, 0,-1,-1, 0,-2,-1,-1,-1,-1,-2,-1, 1, 0,-3,-2, 0,-4, 0,
	-1, 5, 0,-2,-3, 1, 0,-2, 0,-3,-2, 2,-1,-3,-2,-1,-1,-3,-2,-3,-4,-1,
	-2, 0, 6, 1,-3, 0, 0, 0, 1,-3,-3, 0,-2,-3,-2, 1, 0,-4,-2,-3,-4,-1,
	-2,-2, 1, 6,-3, 0, 2,-1,-1,-3,-4,-1,-3,-3,-1, 0,-1,-4,-3,-3,-4,-1,
	 0,-3,-3,-3, 9,-3,-4,-3,-3,-1,-1,-3,-1,-2,-3,-1,-1,-2,-2,-1,-4,-2,
	-1, 1, 0, 0,-3, 5, 2,-2, 0,-3,-2, 1, 0,-3,-1, 0,-1,-2,-1,-2,-4,-1,
	-1, 0, 0, 2,-4, 2, 5,-2, 0,-3,-3, 1,-2,-3,-1, 0,-1,-3,-2,-2,-4,-1,
	 0,-2, 0,-1,-3,-2,-2, 6,-2,-4,-4,-2,-3,-3,-2, 0,-2,-2,-3,-3,-4,-1,
	-2, 0, 1,-1,-3, 0, 0,-2, 8,-3,-3,-1,-2,-1,-2,-1,-2,-2, 2,-3,-4,-1,
	-1,-3,-3,-3,-1,-3,-3,-4,-3, 4, 2,-3, 1, 0,-3,-2,-1,-3,-1, 3,-4,-1,
	-1,-2,-3,-4,-1,-2,-3,-4,-3, 2, 4,-2, 2, 0,-3,-2,-1,-2,-1, 1,-4,-1,
	-1, 2, 0,-1,-3, 1, 1,-2,-1,-3,-2, 5,-1,-3,-1, 0,-1,-3,-2,-2,-4,-1,
	-1,-1,-2,-3,-1, 0,-2,-3,-2, 1, 2,-1, 5, 0,-2,-1,-1,-1,-1, 1,-4,-1,
	-2,-3,-3,-3,-2,-3,-3,-3,-1, 0, 0,-3, 0, 6,-4,-2,-2, 1, 3,-1,-4,-1,
	-1,-2,-2,-1,-3,-1,-1,-2,-2,-3,-3,-1,-2,-4, 7,-1,-1,-4,-3,-2,-4,-2,
	 1,-1, 1, 0,-1, 0, 0, 0,-1,-2,-2, 0,-1,-2,-1, 4, 1,-3,-2,-2,-4, 0,
	 0,-1, 0,-1,-1,-1,-1,-2,-2,-1,-1,-1,-1,-2,-1, 1, 5,-2,-2, 0,-4, 0,
	-3,-3,-4,-4,-2,-2,-3,-2,-2,-3,-2,-3,-1, 1,-4,-3,-2,11, 2,-3,-4,-2,
	-2,-2,-2,-3,-2,-1,-2,-3, 2,-1,-1,-2,-1, 3,-3,-2,-2, 2, 7,-1,-4,-1,
	 0,-3,-3,-3,-1,-2,-2,-3,-3, 3, 1,-2, 1,-1,-2,-2, 0,-3,-1, 4,-4,-1,
	-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4,-4, 1,-4,
	 0,-1,-1,-1,-2,-1,-1,-1,-1,-1,-1,-1,-1,-1,-2, 0, 0,-2,-1,-1,-4,-1
};
namespace mpa {
const int8_t *blosum62() { return ns_mat_blosum62; }

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool timing_on() { const char *e = getenv("MPA_TIMING"); return e && *e && *e != '0'; }   // (read per call: the tests flip it)
void timing_note(const char *what, double ms) { if (timing_on()) fprintf(stderr, "[mpa-timing] %-28s %9.3f ms\n", what, ms); }

// CPU accounting (MPA_TIMING=1): the box gives a process a CPU quota, not the cores it sees, so what a pipeline stage COSTS is its
// thread CPU time (CLOCK_THREAD_CPUTIME_ID: spinning in a wait counts, sleeping does not), not the wall time of its threads
double thread_cpu_ms()
{
	struct timespec ts;
	clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
static std::mutex g_cpu_mu;
static std::vector<std::pair<std::string, double>> g_cpu_acc;
void cpu_add(const char *label, double ms)
{
	std::lock_guard<std::mutex> g(g_cpu_mu);
	for (auto &e : g_cpu_acc) if (e.first == label) { e.second += ms; return; }
	g_cpu_acc.emplace_back(label, ms);
}
void cpu_report_and_reset(const char *header, double wall_ms)
{
	std::lock_guard<std::mutex> g(g_cpu_mu);
	double tot = 0;
	for (auto &e : g_cpu_acc) tot += e.second;
	if (timing_on()) {
		fprintf(stderr, "[mpa-cpu] %s: wall %.1f ms, accounted thread CPU %.1f ms (%.1f cores)\n", header, wall_ms, tot, wall_ms > 0 ? tot / wall_ms : 0.0);
		for (auto &e : g_cpu_acc) fprintf(stderr, "[mpa-cpu]   %-44s %10.1f ms  %5.1f %%\n", e.first.c_str(), e.second, tot > 0 ? 100 * e.second / tot : 0.0);
	}
	g_cpu_acc.clear();
}

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

} // namespace mpa

using namespace mpa;

// The DP executor keeps up to 16 kernels of a round on separate streams, two rounds at a time in a stream of batches; HIP's
// default of 4 hardware queues would serialise them.  The runtime reads the variable when it initialises, so it is set when the
// library is loaded, unless the host application has chosen a value itself.
__attribute__((constructor)) static void mpa_default_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" {

const char *mpa_last_error(void) { return g_err.c_str(); }
const char *mpa_version(void) { return "mpamd-0.1 (miniprot 0.18-r281 compatible)"; }
void mpa_free(void *p) { free(p); }

void mpa_idxopt_init(mpa_idxopt_t *io)
{
	memset(io, 0, sizeof(*io));
	io->trans_code = 1, io->bbit = 8, io->min_aa_len = 30, io->kmer = 6, io->mod_bit = 1;
}

// the stop row/column of the matrix carries the frameshift penalty (nasw-tab.c:149-156)
static void set_stop_score(int8_t *mat, int32_t pen)
{
	const int stop = 20;
	int8_t keep = mat[stop * 22 + stop];
	for (int i = 0; i < 22; ++i) mat[stop * 22 + i] = mat[i * 22 + stop] = (int8_t)-pen;
	mat[stop * 22 + stop] = keep;
}

void mpa_mapopt_set_fs(mpa_mapopt_t *mo, int32_t fs) { mo->fs = fs; set_stop_score(mo->mat, fs); }

void mpa_mapopt_set_max_intron(mpa_mapopt_t *mo, int64_t gsize)
{
	int64_t x = (int64_t)(sqrt((double)gsize) * 3.6 + 1.);    // options.c:8,34
	if (x < mo->min_max_intron) x = mo->min_max_intron;
	if (x > mo->max_max_intron) x = mo->max_max_intron;
	mo->bw = mo->max_intron = (int32_t)x;
}

void mpa_mapopt_init(mpa_mapopt_t *mo)
{
	memset(mo, 0, sizeof(*mo));
	mo->mini_batch_size = 2000000;
	mo->max_occ = 20000, mo->max_gap = 1000, mo->max_intron = 200000;
	mo->min_max_intron = 10000, mo->max_max_intron = 300000;
	mo->bw = mo->max_intron;
	mo->min_chn_cnt = 3, mo->max_chn_max_skip = 25, mo->max_chn_iter = 1000000, mo->min_chn_sc = 0;
	mo->chn_coef_log = 0.75f;
	mo->max_ext = 10000, mo->max_ava = 1000;
	mo->mask_level = 0.5f, mo->mask_len = INT32_MAX;
	mo->pri_ratio = 0.7f, mo->best_n = 30, mo->out_n = 1000, mo->out_sim = 0.99f, mo->out_cov = 0.1f;
	mo->kmer2 = 5;
	mo->go = 11, mo->ge = 1, mo->io = 29, mo->fs = 23, mo->io_end = 19;
	mo->ie_coef = .5f;
	mo->sp_model = 1, mo->sp_null_bonus = -7, mo->sp_max_bonus = 14, mo->sp_scale = 1.0f;
	mo->end_bonus = 5, mo->xdrop = 100, mo->asize = 22;
	memcpy(mo->mat, blosum62(), 484);
	set_stop_score(mo->mat, mo->fs);
	mo->gff_delim = -1, mo->gff_prefix = "MP", mo->max_intron_flank = 200;
}

// mp_map2ns_opt (align.c:50-60) with ns_opt_set_sp (nasw-tab.c:124-129) folded in
void mpa_dpopt_from_mapopt(const mpa_mapopt_t *mo, mpa_dpopt_t *dp)
{
	static const int32_t models[3][6] = { {0,0,0,0,0,0}, {8,15,21,30,0,0}, {8,15,21,30,4,4} };
	int m = mo->sp_model < 0 || mo->sp_model > 2 ? 0 : mo->sp_model;
	dp->go = mo->go, dp->ge = mo->ge, dp->fs = mo->fs, dp->xdrop = mo->xdrop, dp->end_bonus = mo->end_bonus;
	dp->ie_coef = mo->ie_coef, dp->sp_null_bonus = mo->sp_null_bonus;
	for (int i = 0; i < 6; ++i) dp->sp[i] = (int32_t)(models[m][i] * mo->sp_scale + .499f);
	memcpy(dp->mat, mo->mat, 484);
}

} // extern "C"
