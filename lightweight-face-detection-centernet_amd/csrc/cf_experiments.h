/* Entry points that exist only in an experiments build of the library (make EXP=1 -> libcenterface_hip_exp.so, -DCF_EXPERIMENTS):
 * measured and rejected designs kept for A/B runs.  Not part of the product ABI (include/centerface_hip.h). */
#pragma once
#include "centerface_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Two-lane schedule over two contexts that alternate batches (experimental): enqueues the VALU-bound
 * front of the new batch on `cur` alone, then the back half (wide late blocks, neck, heads) of the batch pending on `prev`
 * underneath the mid-size blocks of the new batch.  `prev` may be NULL (first batch).  After the call prev's results can be
 * decoded (cf_decode_topk* / cf_gather_topk on prev); cur has no decodable result until its own back half has been launched by
 * the NEXT cf_forward_lanes(other, cur, ...) or by cf_forward_lanes_flush(cur).  Same arithmetic as cf_forward. */
int cf_forward_lanes(cf_ctx* cur, cf_ctx* prev, const void* in, int in_format, int in_on_device, int B);
int cf_forward_lanes_flush(cf_ctx* ctx);
#ifdef __cplusplus
}
#endif
