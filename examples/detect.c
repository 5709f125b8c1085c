/* Plain-C use of the drop-in boundary (include/centerface_hip.h): what a cgo / JNI / N-API stub would call.
 *
 *   gcc -std=c99 -Iinclude examples/detect.c -o detect \
 *       -Llightweight-face-detection-centernet_amd -lcenterface_hip -Wl,-rpath,$PWD/lightweight-face-detection-centernet_amd
 *   ./detect weights.bin 640 640 4
 *
 * weights.bin is the flat tensor file written by tools/export_weights.py: for each of the 94 tensors of the
 * reference's state_dict, in order: name (64 bytes, NUL padded), int32 dtype (0 = f32, 1 = i64), int32 ndim,
 * int64 dims[4], then the raw data.  The program builds a random uint8 BGR batch, runs forward + top-K decode
 * and prints the best detections; exit code 0 on success.  No C++ and no Python anywhere on this path. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "centerface_hip.h"

#define CHECK(ctx, call)                                                                                   \
    do { int rc_ = (call); if (rc_ != CF_OK) { fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_,          \
         cf_strerror(rc_), cf_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s weights.bin H W batch\n", argv[0]); return 2; }
    const int H = atoi(argv[2]), W = atoi(argv[3]), B = atoi(argv[4]), K = 10;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    enum { MAXT = 128 };
    static cf_tensor_desc descs[MAXT];
    static char names[MAXT][64];
    int n = 0;
    for (; n < MAXT; ++n) {
        int32_t dtype, ndim; int64_t dims[4];
        if (fread(names[n], 1, 64, f) != 64) break;
        if (fread(&dtype, 4, 1, f) != 1 || fread(&ndim, 4, 1, f) != 1 || fread(dims, 8, 4, f) != 4) return 2;
        size_t count = 1;
        for (int i = 0; i < ndim; ++i) count *= (size_t)dims[i];
        const size_t bytes = count * (dtype == 1 ? 8 : 4);
        void* data = malloc(bytes ? bytes : 1);
        if (fread(data, 1, bytes, f) != bytes) return 2;
        memset(&descs[n], 0, sizeof descs[n]);
        descs[n].name = names[n]; descs[n].data = data; descs[n].ndim = ndim; descs[n].dtype = dtype;
        for (int i = 0; i < ndim; ++i) descs[n].dims[i] = dims[i];
    }
    fclose(f);

    cf_ctx* ctx = NULL;
    CHECK(NULL, cf_create(0, B, H, W, CF_BF16, CF_FLAG_COLLAPSE_HEADS, &ctx));
    CHECK(ctx, cf_load_weights(ctx, descs, n));

    uint8_t* img = (uint8_t*)malloc((size_t)B * H * W * 3);
    uint32_t s = 12345u;
    for (size_t i = 0; i < (size_t)B * H * W * 3; ++i) { s = s * 1664525u + 1013904223u; img[i] = (uint8_t)(s >> 24); }
    float* dets = (float*)malloc((size_t)B * K * 6 * sizeof(float));
    float* lms = (float*)malloc((size_t)B * K * 10 * sizeof(float));
    int64_t* inds = (int64_t*)malloc((size_t)B * K * sizeof(int64_t));
    CHECK(ctx, cf_forward(ctx, img, CF_IN_U8_HWC_BGR, 0, B));
    CHECK(ctx, cf_decode_topk(ctx, K, 1, dets, lms, inds, 0));
    for (int b = 0; b < B; ++b)
        printf("image %d: best score %.4f at cell %lld, box [%.2f %.2f %.2f %.2f] (map units)\n", b, dets[(b * K) * 6 + 4],
               (long long)inds[b * K], dets[(b * K) * 6], dets[(b * K) * 6 + 1], dets[(b * K) * 6 + 2], dets[(b * K) * 6 + 3]);
    /* Multi-GPU (one process or thread per GPU, batch sharded by rank): the only exchange is the all-gather of the final
     * records.  Rank 0 creates the 128-byte id and ships it to the other ranks by any channel (file, socket, MPI ...);
     * every rank then calls cf_comm_create(ctx, rank, world, id, &comm) and, after each cf_forward on its shard,
     * cf_gather_topk, which returns the records of ALL ranks in rank-major (= unsharded batch) order.  Shown here with
     * world = 1 (`./detect weights.bin H W batch gather`). */
    if (argc > 5 && strcmp(argv[5], "gather") == 0) {
        char id[CF_COMM_ID_BYTES];
        cf_comm* comm = NULL;
        const int world = 1, rank = 0;
        float* rec = (float*)malloc((size_t)world * B * K * 16 * sizeof(float));
        CHECK(NULL, cf_comm_unique_id(id, (int)sizeof id));
        CHECK(ctx, cf_comm_create(ctx, rank, world, id, &comm));
        /* declare the shard (B images x K records) once: the only extra collective of the gather path, enqueued and polled --
         * a host with a deadline gives up with cf_comm_abort instead of blocking in the first gather */
        CHECK(ctx, cf_comm_set_shard(comm, B, K));
        { int q; while ((q = cf_comm_query(comm)) == 1) { } if (q != 0) { fprintf(stderr, "shard agreement: %s\n", cf_comm_last_error(comm)); return 1; } }
        CHECK(ctx, cf_forward(ctx, img, CF_IN_U8_HWC_BGR, 0, B));
        CHECK(ctx, cf_gather_topk(ctx, comm, K, 1, rec, 0));        /* one ncclAllGather: [header | records] per rank */
        for (int b = 0; b < world * B; ++b)
            if (rec[(size_t)b * K * 16 + 4] != dets[(size_t)b * K * 6 + 4]) { fprintf(stderr, "gathered record differs\n"); return 1; }
        printf("gathered %d x %d records over RCCL (world %d)\n", world * B, K, world);
        /* a host with a deadline polls instead of blocking: 0 = every enqueued gather has completed */
        if (cf_comm_query(comm) != 0) { fprintf(stderr, "gather still running after a blocking gather\n"); return 1; }
        CHECK(ctx, cf_comm_destroy(comm));
        /* A single process / thread that owns one context per GPU creates all communicators in ONE grouped call
         * (un-grouped, the first ncclCommInitRank would wait forever for the others); ONE communicator per rank, shared
         * by all contexts of that rank.  Shown with n = 1; cf_comm_abort is the way out of a collective that hangs. */
        {
            cf_ctx* ctxs[1]; cf_comm* comms[1];
            ctxs[0] = ctx;
            CHECK(ctx, cf_comm_create_all(ctxs, 1, comms));
            CHECK(ctx, cf_forward(ctx, img, CF_IN_U8_HWC_BGR, 0, B));
            CHECK(ctx, cf_gather_topk(ctx, comms[0], K, 1, rec, 0));
            for (int b = 0; b < B; ++b)
                if (rec[(size_t)b * K * 16 + 4] != dets[(size_t)b * K * 6 + 4]) { fprintf(stderr, "gathered record differs (grouped)\n"); return 1; }
            printf("grouped communicator: gathered %d x %d records\n", B, K);
            CHECK(ctx, cf_comm_abort(comms[0]));
        }
        free(rec);
    }
    /* Dry run of an N-rank gather on ONE GPU (`./detect weights.bin H W batch dryrun N`): no RCCL -- a loopback communicator lets this
     * process play the N ranks in turn.  The batch is dealt to the ranks image by image (shards of one image: B ranks at most); every
     * rank decodes its shard, writes its slot header and deposits the slot; the call of the last rank runs the header check and the
     * rank-major unpack of the real gather.  The gathered records must equal the unsharded decode above, image for image: what a
     * deployment checks about its shard math before the multi-GPU node exists. */
    if (argc > 6 && strcmp(argv[5], "dryrun") == 0) {
        const int world = atoi(argv[6]);
        if (world < 1 || world > B) { fprintf(stderr, "dryrun: world must be in [1, batch]\n"); return 2; }
        cf_comm* comm = NULL;
        float* rec = (float*)malloc((size_t)world * K * 16 * sizeof(float));
        CHECK(ctx, cf_comm_create_loopback(ctx, world, &comm));
        CHECK(ctx, cf_comm_set_shard(comm, 1, K));
        for (int r = 0; r < world; ++r) {
            CHECK(ctx, cf_forward(ctx, img + (size_t)r * H * W * 3, CF_IN_U8_HWC_BGR, 0, 1));       /* rank r's shard: image r */
            if (cf_comm_loopback_rank(comm, r) != CF_OK) { fprintf(stderr, "cf_comm_loopback_rank failed\n"); return 1; }
            CHECK(ctx, cf_gather_topk(ctx, comm, K, 1, rec, 0));                                    /* filled by the LAST rank's call */
        }
        for (int b = 0; b < world; ++b)
            for (int k = 0; k < K; ++k)
                for (int e = 0; e < 5; ++e)
                    if (rec[((size_t)b * K + k) * 16 + e] != dets[((size_t)b * K + k) * 6 + e]) { fprintf(stderr, "dry run: record %d of image %d differs from the unsharded decode\n", k, b); return 1; }
        printf("dry run: %d ranks x 1 image gathered in rank-major order = the unsharded decode\n", world);
        CHECK(ctx, cf_comm_destroy(comm));
        free(rec);
    }
    CHECK(ctx, cf_destroy(ctx));
    return 0;
}
