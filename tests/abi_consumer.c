/*
 * abi_consumer.c -- a plain C99 caller of libcmax_hip.so, compiled against include/cmax_hip.h ALONE.
 *
 * The Python package reaches the library through its own ctypes table (event_based_optical_flow_amd/_lib.py); this
 * program is the other kind of consumer the header is written for: no Python, no torch, only the HIP runtime for device
 * memory.  tests/test_abi_consumer.py builds it (gcc, -std=c99 -pedantic), runs it on a golden case and compares the
 * numbers it prints with the reference's values.
 *
 *   abi_consumer <input.bin> [n_evaluations]
 *
 * input.bin (little endian): int32 H, W, n, model (CMAX_MODEL_*), cost (CMAX_COST_*), n_motion; double sigma;
 *                            double events[n][4]; float motion[n_motion]
 * stdout: "loss <v>", "grad <i> <v>" ... (the first 16 and the last gradient entries), "gradsum <v>", "iwesum <v>"
 * exit code: 0 ok, 1 usage / IO, 2 a cmax_* call failed (message on stderr), 3 a HIP call failed.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "cmax_hip.h"

#define HIP_OK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));             \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

#define CMAX_OK(call)                                                                     \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != 0) {                                                                   \
            fprintf(stderr, "%s returned %d: %s\n", #call, rc_, cmax_last_error());       \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char **argv) {
    int32_t hdr[6];
    double sigma = 0.0;
    double *events_host = NULL, *events_dev = NULL, *result_dev = NULL;
    float *motion_host = NULL, *motion_dev = NULL, *iwe_dev = NULL, *iwe_host = NULL;
    void *grad_dev = NULL;
    double result[8];
    int64_t n_packed = 0, n_dropped = 0;
    int n_eval = 1, k, H, W, n, model, cost, n_motion;
    size_t grad_bytes, i;
    cmax_handle_t h = NULL;
    cmax_objective_t desc;
    hipStream_t stream;
    FILE *f;

    if (argc < 2) {
        fprintf(stderr, "usage: %s input.bin [n_evaluations]\n", argv[0]);
        return 1;
    }
    if (argc > 2) n_eval = atoi(argv[2]);
    f = fopen(argv[1], "rb");
    if (!f || fread(hdr, sizeof(int32_t), 6, f) != 6 || fread(&sigma, sizeof(double), 1, f) != 1) {
        fprintf(stderr, "cannot read the header of %s\n", argv[1]);
        return 1;
    }
    H = hdr[0], W = hdr[1], n = hdr[2], model = hdr[3], cost = hdr[4], n_motion = hdr[5];
    events_host = (double *)malloc((size_t)n * 4 * sizeof(double));
    motion_host = (float *)malloc((size_t)n_motion * sizeof(float));
    if (!events_host || !motion_host || fread(events_host, sizeof(double), (size_t)n * 4, f) != (size_t)n * 4 ||
        fread(motion_host, sizeof(float), (size_t)n_motion, f) != (size_t)n_motion) {
        fprintf(stderr, "cannot read the payload of %s\n", argv[1]);
        return 1;
    }
    fclose(f);

    if (cmax_abi_version() != CMAX_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header ABI %d\n", cmax_abi_version(), CMAX_ABI_VERSION);
        return 2;
    }
    if (cmax_sizeof_objective() != (int)sizeof(cmax_objective_t)) {
        fprintf(stderr, "cmax_objective_t: library %d bytes, this compiler %d\n", cmax_sizeof_objective(), (int)sizeof(cmax_objective_t));
        return 2;
    }

    HIP_OK(hipSetDevice(0));
    HIP_OK(hipStreamCreate(&stream));
    HIP_OK(hipMalloc((void **)&events_dev, (size_t)n * 4 * sizeof(double)));
    HIP_OK(hipMalloc((void **)&motion_dev, (size_t)n_motion * sizeof(float)));
    HIP_OK(hipMalloc((void **)&result_dev, 8 * sizeof(double)));
    HIP_OK(hipMalloc((void **)&iwe_dev, (size_t)H * W * sizeof(float)));
    grad_bytes = model == CMAX_MODEL_2DOF ? 2 * sizeof(double) : (size_t)n_motion * sizeof(float);
    HIP_OK(hipMalloc(&grad_dev, grad_bytes));
    HIP_OK(hipMemcpyAsync(events_dev, events_host, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(motion_dev, motion_host, (size_t)n_motion * sizeof(float), hipMemcpyHostToDevice, stream));

    /* the sequence a solver written in C would run: one handle per sensor, one set_events per batch, many objectives */
    CMAX_OK(cmax_create(H, W, 0, 0, &h));
    CMAX_OK(cmax_set_events(h, events_dev, CMAX_F64, n, 0, 0.0, 0.0, 0, (cmax_stream_t)stream));
    CMAX_OK(cmax_batch_info(h, &n_packed, &n_dropped, NULL, NULL));

    memset(&desc, 0, sizeof(desc));
    desc.model = model;
    desc.cost = cost;
    desc.normalized = 0;
    desc.minimize = 1; /* direction "minimize": loss = -contrast (src/costs/image_variance.py:56-58) */
    desc.omit_boundary = 1;
    desc.normalize_t = 1;
    desc.n_ref = 1;
    desc.ref_mode[0] = CMAX_REF_FIRST;
    desc.mult[0] = 1.0;
    desc.sigma = sigma;
    for (k = 0; k < n_eval; ++k)
        CMAX_OK(cmax_objective(h, &desc, motion_dev, result_dev, grad_dev, (cmax_stream_t)stream));
    CMAX_OK(cmax_copy_iwe(h, 0, iwe_dev, (cmax_stream_t)stream));
    HIP_OK(hipMemcpyAsync(result, result_dev, sizeof(result), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    printf("packed %lld dropped %lld\n", (long long)n_packed, (long long)n_dropped);
    printf("loss %.17g\n", result[0]);
    printf("contrast %.17g\n", result[1]);
    if (model == CMAX_MODEL_2DOF) {
        double g[2];
        HIP_OK(hipMemcpy(g, grad_dev, sizeof(g), hipMemcpyDeviceToHost));
        printf("grad 0 %.17g\ngrad 1 %.17g\n", g[0], g[1]);
        printf("gradsum %.17g\n", g[0] + g[1]);
    } else {
        float *g = (float *)malloc(grad_bytes);
        double s = 0.0;
        if (!g) return 1;
        HIP_OK(hipMemcpy(g, grad_dev, grad_bytes, hipMemcpyDeviceToHost));
        for (i = 0; i < (size_t)n_motion; ++i) s += (double)g[i];
        for (i = 0; i < 16 && i < (size_t)n_motion; ++i) printf("grad %d %.9g\n", (int)i, (double)g[i]);
        printf("gradsum %.17g\n", s);
        free(g);
    }
    iwe_host = (float *)malloc((size_t)H * W * sizeof(float));
    if (!iwe_host) return 1;
    HIP_OK(hipMemcpy(iwe_host, iwe_dev, (size_t)H * W * sizeof(float), hipMemcpyDeviceToHost));
    {
        double s = 0.0;
        for (i = 0; i < (size_t)H * W; ++i) s += (double)iwe_host[i];
        printf("iwesum %.17g\n", s);
    }

    CMAX_OK(cmax_destroy(h));
    /* error convention: bad arguments come back as negative codes with a message, nothing crashes */
    if (cmax_create(0, 0, 0, 0, &h) != CMAX_EINVAL || cmax_last_error()[0] == '\0') {
        fprintf(stderr, "cmax_create(0, 0) should have returned CMAX_EINVAL with a message\n");
        return 2;
    }
    printf("einval_ok 1\n");
    (void)hipFree(events_dev);
    (void)hipFree(motion_dev);
    (void)hipFree(result_dev);
    (void)hipFree(iwe_dev);
    (void)hipFree(grad_dev);
    (void)hipStreamDestroy(stream);
    free(events_host);
    free(motion_host);
    free(iwe_host);
    return 0;
}
