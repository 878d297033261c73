// gs32_exec.hip -- ns_global_gs32b on the device (included by dp_exec.hip).  One wavefront per call runs gs32_core.h's sweep
// (policy CoopWave); the per-row records, the query profile and the traceback walk are the int16 path's own kernels
// (k_prep_rows, k_prep_prof, k_backtrack: the record and traceback-word formats are the same).  Not a throughput path: nothing
// in miniprot calls the 32-bit operator (see gs32_core.h); a batch of calls still runs one wave per call side by side.
#include "gs32_core.h"

namespace mpa {

__global__ __launch_bounds__(64) void k_gs32(const DTask *tasks, int32_t n, const uint32_t *rec, const int16_t *prof, int32_t *scratch, uint16_t *tbpool,
                                             int32_t *score, Gs32Params p)
{
	const int32_t k = blockIdx.x;
	if (k >= n) return;
	const DTask t = tasks[k];
	Gs32Params pk = p;
	pk.io = t.io;
	const bool want_tb = (t.flag & 1) && !(t.flag & 6);                  // NS_F_CIGAR and no extension flag (NS_GEN_PREPARE, nasw-sse.c:248-249)
	const int32_t sc = gs32_sweep<CoopWave>(pk, t.nl, t.al, rec + t.rec_off, prof + t.prof_off, t.pw, gs32_carve(scratch + t.bnd_off, t.al),
	                                        want_tb ? tbpool + t.tb_off : nullptr, t.ncol);
	if (threadIdx.x == 0) score[k] = sc;
}

} // namespace mpa

extern "C" int mpa_dp_run32(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_dpopt_t *opt, const mpa_qbatch_t *q,
                            int64_t n, const mpa_dp_task_t *in, mpa_dp_rst_t *rst, uint32_t **cigar_pool, int64_t *n_pool)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&]() -> int {
		if (cigar_pool) *cigar_pool = nullptr;
		if (n_pool) *n_pool = 0;
		if (!ctx) { set_error("no device context"); return MPA_ERR_NO_DEVICE; }
		if (ctx->device >= mpa_idx_s::kMaxDevices || !mi->dev[ctx->device]) { set_error("index is not resident on this context's device (call mpa_idx_to_device)"); return MPA_ERR_ARG; }
		if (n <= 0) return MPA_OK;
		if (n > (1 << 20)) { set_error("mpa_dp_run32: at most 2^20 calls per batch"); return MPA_ERR_ARG; }
		// the sweep's arithmetic wraps like the reference's _mm_add/_sub_epi32; these bounds keep everything far from wrapping
		// (|H| <= 2^30 + al * 127 + nl * 2^20) and make the closed form of the gap chain valid (go >= 0)
		if (opt->go < 0 || opt->go > (1 << 20) || opt->ge < 0 || opt->ge > (1 << 20) || opt->fs < 0 || opt->fs > (1 << 20)) {
			set_error("32-bit DP parameters outside the supported range (0 <= go, ge, fs <= 2^20)"); return MPA_ERR_UNSUPPORTED;
		}
		const int32_t wide_ge = opt->ge > 255 || opt->fs > 255;   // the records then carry a stop flag in place of the row's gap extension
		HIP_TRY(hipSetDevice(ctx->device));
		hipStream_t s = ctx->stream;
		std::vector<DTask> T((size_t)n);
		std::vector<PrepChunk> chunks;
		std::vector<int32_t> walk;
		int64_t rec_total = 0, prof_total = 0, tb_total = 0, cig_total = 0, scr_total = 0;
		int32_t max_nl = 0;
		for (int64_t k = 0; k < n; ++k) {
			const mpa_dp_task_t &x = in[k];
			DTask &t = T[k];
			memset(&t, 0, sizeof(t));
			if (x.nl < 0 || x.al <= 0 || x.qid < 0 || x.qid >= q->n_seq || x.io < 0 || x.io > (1 << 20)) { set_error("malformed DP task"); return MPA_ERR_ARG; }
			if (x.vid < 0 || x.vid >= 2 * (int32_t)mi->ctg.size() || x.nt_off < 0 || x.nt_off + (int64_t)x.nl > mi->ctg[x.vid >> 1].len ||
			    x.aa_off < 0 || (int64_t)x.aa_off + x.al > q->q_off[x.qid + 1] - q->q_off[x.qid]) {
				set_error("DP task " + std::to_string(k) + " reaches outside its contig or its query"); return MPA_ERR_ARG;
			}
			t.nt_off = x.nt_off, t.vid = x.vid, t.nl = x.nl, t.al = x.al, t.flag = x.flag, t.io = x.io;
			t.q_off = q->q_off[x.qid] + x.aa_off - q->q_off[0];
			t.ncol = (x.al + 7) / 8 * 8, t.pw = t.ncol, t.out_idx = (int32_t)k;
			t.rec_off = rec_total, rec_total += t.nl;
			t.prof_off = prof_total, prof_total += (int64_t)22 * t.pw;
			t.bnd_off = scr_total, scr_total += gs32_scratch_words(t.al);     // (int32 words of the sweep's scratch)
			max_nl = std::max(max_nl, t.nl);
			for (int32_t r = 0; r < t.nl; r += MPA_PREP_CHUNK_ROWS) chunks.push_back(PrepChunk{ (int32_t)k, r });
			if ((x.flag & MPA_F_CIGAR) && !(x.flag & (MPA_F_EXT_LEFT | MPA_F_EXT_RIGHT))) {
				t.tb_off = tb_total, tb_total += (int64_t)t.nl * t.ncol;
				t.cig_cap = t.nl + t.al + 4, t.cig_off = cig_total, cig_total += t.cig_cap;
				walk.push_back((int32_t)k);
			}
		}
		if ((uint64_t)tb_total * 2 > ctx->tb_budget) { set_error("mpa_dp_run32: traceback matrices of this batch exceed the context's traceback budget; pass fewer calls"); return MPA_ERR_UNSUPPORTED; }
		rec_total += max_nl + 96;
		const int64_t q_bytes = q->q_off[q->n_seq] - q->q_off[0];
		int rc;
		if ((rc = ctx->tasks.ensure(sizeof(DTask) * (size_t)n)) || (rc = ctx->chunks.ensure(sizeof(PrepChunk) * chunks.size() + 16)) ||
		    (rc = ctx->qseq.ensure((size_t)q_bytes + 16)) || (rc = ctx->rec.ensure((size_t)rec_total * 4)) || (rc = ctx->prof.ensure((size_t)prof_total * 2 + 16)) ||
		    (rc = ctx->bnd.ensure((size_t)scr_total * 4 + 16)) || (rc = ctx->tb.ensure((size_t)tb_total * 2 + 16)) || (rc = ctx->cig.ensure((size_t)cig_total * 4 + 16)) ||
		    (rc = ctx->ncig.ensure((size_t)n * 4)) || (rc = ctx->score.ensure((size_t)n * 4)) || (rc = ctx->list.ensure(walk.size() * 4 + 16)))
			return rc;
		HIP_TRY(hipMemcpyAsync(ctx->tasks.p, T.data(), sizeof(DTask) * (size_t)n, hipMemcpyHostToDevice, s));
		if (!chunks.empty()) HIP_TRY(hipMemcpyAsync(ctx->chunks.p, chunks.data(), sizeof(PrepChunk) * chunks.size(), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(ctx->qseq.p, q->seqs + q->q_off[0], (size_t)q_bytes, hipMemcpyHostToDevice, s));
		if (!walk.empty()) HIP_TRY(hipMemcpyAsync(ctx->list.p, walk.data(), walk.size() * 4, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemsetAsync(ctx->ncig.p, 0, (size_t)n * 4, s));
		HIP_TRY(hipStreamSynchronize(s));                                      // (the sources above are pageable vectors of this call)
		DevTables tabs;
		memcpy(tabs.aa20, tab_aa20(), 256);
		memcpy(tabs.codon, tab_codon(), 64);
		memcpy(tabs.mat, opt->mat, 484);
		DpConst dc;
		dc.go = opt->go, dc.ge = opt->ge, dc.fs = opt->fs, dc.xdrop = opt->xdrop, dc.end_bonus = opt->end_bonus;
		for (int k = 0; k < 6; ++k) dc.sp[k] = opt->sp[k];
		dc.sp_null_bonus = opt->sp_null_bonus;
		dc.wide_ge = wide_ge;
		DevGenome dg{ mi->dev[ctx->device]->seq, mi->dev[ctx->device]->ctg_off, mi->dev[ctx->device]->ctg_len, mi->dev[ctx->device]->spsc, mi->l_seq };
		if (!chunks.empty())
			hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)chunks.size()), dim3(256), 0, s, dg, ctx->tasks.as<DTask>(), ctx->chunks.as<PrepChunk>(), ctx->rec.as<uint32_t>(), dc, tabs);
		hipLaunchKernelGGL(k_prep_prof, dim3((unsigned)n), dim3(256), 0, s, ctx->tasks.as<DTask>(), ctx->qseq.as<char>(), ctx->prof.as<int16_t>(), tabs);
		hipLaunchKernelGGL(k_gs32, dim3((unsigned)n), dim3(64), 0, s, ctx->tasks.as<DTask>(), (int32_t)n, ctx->rec.as<uint32_t>(), ctx->prof.as<int16_t>(),
		                   ctx->bnd.as<int32_t>(), ctx->tb.as<uint16_t>(), ctx->score.as<int32_t>(), Gs32Params{ opt->go, opt->ge, 0, opt->fs, wide_ge });
		if (!walk.empty())
			hipLaunchKernelGGL(k_backtrack, dim3((unsigned)walk.size()), dim3(64), 0, s, ctx->tasks.as<DTask>(), ctx->list.as<int32_t>(), (int32_t)walk.size(),
			                   ctx->tb.as<uint16_t>(), ctx->cig.as<uint32_t>(), ctx->ncig.as<int32_t>());
		HIP_TRY(hipGetLastError());
		std::vector<int32_t> sc((size_t)n), nc((size_t)n);
		std::vector<uint32_t> cig((size_t)cig_total + 1);
		HIP_TRY(hipMemcpyAsync(sc.data(), ctx->score.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(nc.data(), ctx->ncig.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
		if (cig_total) HIP_TRY(hipMemcpyAsync(cig.data(), ctx->cig.p, (size_t)cig_total * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(wait_stream(ctx, s));
		int64_t pool_n = 0;
		for (int32_t k : walk) pool_n += nc[k];
		uint32_t *pool = (uint32_t*)malloc((size_t)(pool_n > 0 ? pool_n : 1) * 4);
		if (!pool) { set_error("out of memory"); return MPA_ERR_HIP; }
		int64_t at = 0;
		for (int64_t k = 0; k < n; ++k) {
			rst[k].nt_len = T[k].nl, rst[k].aa_len = T[k].al, rst[k].score = sc[k], rst[k].n_cigar = 0, rst[k].cigar_off = 0;   // (ns_global_gs32b has no extension mode: lengths as given)
			if (T[k].cig_cap > 0 && nc[k] > 0) {
				memcpy(pool + at, cig.data() + T[k].cig_off, (size_t)nc[k] * 4);
				rst[k].n_cigar = nc[k], rst[k].cigar_off = at, at += nc[k];
			}
		}
		if (cigar_pool) *cigar_pool = pool; else free(pool);
		if (n_pool) *n_pool = pool_n;
		return MPA_OK;
	});
}
