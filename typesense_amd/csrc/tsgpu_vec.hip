// TEMPORARY stubs (vector path lands next)
#include "tsgpu_host.h"
using namespace tsgpu;
extern "C" {
void tsgpu_vec_destroy_all(tsgpu_ctx*) {}
uint64_t tsgpu_vec_device_bytes(tsgpu_ctx*) { return 0; }
int tsgpu_vec_create(tsgpu_ctx*, uint32_t, uint32_t, int, uint64_t) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_vec_upsert(tsgpu_ctx*, uint32_t, const uint64_t*, const float*, uint32_t, int) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_vec_delete(tsgpu_ctx*, uint32_t, uint64_t) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_vec_get(tsgpu_ctx*, uint32_t, uint64_t, float*) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
uint64_t tsgpu_vec_count(tsgpu_ctx*, uint32_t) { return 0; }
int tsgpu_vec_knn_batch(tsgpu_ctx*, uint32_t, const float*, int, uint32_t, uint32_t, const uint32_t*, uint32_t, const uint32_t*, uint32_t, float*, uint64_t*, uint32_t*, int) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_vec_distances(tsgpu_ctx*, uint32_t, const float*, const uint64_t*, uint32_t, float*) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_vector_search_batch(tsgpu_ctx*, uint32_t, const tsgpu_vec_query*, const float*, int, uint32_t, tsgpu_hits*) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_hybrid_search_batch(tsgpu_ctx*, const tsgpu_kw_query*, uint32_t, const tsgpu_hybrid_params*, const float*, int, uint32_t, tsgpu_hits*) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
int tsgpu_merge_shard_hits(const tsgpu_hits*, const uint64_t*, uint32_t, uint32_t, uint32_t, tsgpu_hits*) { return fail(TSGPU_ERR_UNSUPPORTED, "nyi"); }
}
