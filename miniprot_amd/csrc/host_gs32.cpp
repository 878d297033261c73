// host_gs32.cpp -- the HOST instance of gs32_core.h's sweep, exported as a diagnostic.  It is a test model of the device
// code (the same source compiled for a team of one): tests/test_gs32_model.py pins it, traceback words and all, against the
// reference's ns_global_gs32b.  No product path calls it -- ns_global_gs32b / mpa_dp_run32 run k_gs32 on the GPU or fail.
#include <cstdlib>
#include <vector>
#include "../../include/mpamd.h"
#include "gs32_core.h"

extern "C" int32_t mpa_dbg_gs32_model(int32_t nl, int32_t al, const uint32_t *rec, const int16_t *prof, int32_t pw, int32_t go, int32_t ge, int32_t io, int32_t fs,
                                      uint16_t *tb, int32_t ncol)
{
	if (al <= 0 || nl < 0 || go < 0) return GS32_NEG;
	std::vector<int32_t> scratch((size_t)mpa::gs32_scratch_words(al) + 4);
	return mpa::gs32_sweep<mpa::CoopSerial>(mpa::Gs32Params{ go, ge, io, fs }, nl, al, rec, prof, pw, mpa::gs32_carve(scratch.data(), al), tb, ncol);
}
