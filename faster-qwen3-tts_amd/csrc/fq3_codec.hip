// 12 Hz codec decoder entry points (placeholder until csrc/codec kernels land in this round).
#include "../../include/fq3hip.h"
extern "C" int fq3_codec_create(const fq3_codec_config*, fq3_codec**) { return FQ3_EUNSUPPORTED; }
extern "C" int fq3_codec_destroy(fq3_codec*) { return FQ3_OK; }
extern "C" int fq3_codec_bind(fq3_codec*, const char*, const void*, int64_t) { return FQ3_EUNSUPPORTED; }
extern "C" int fq3_codec_finalize(fq3_codec*, void*) { return FQ3_EUNSUPPORTED; }
extern "C" int64_t fq3_codec_num_samples(const fq3_codec*, int) { return -1; }
extern "C" int fq3_codec_decode(fq3_codec*, const int64_t*, int, float*, void*) { return FQ3_EUNSUPPORTED; }
