/* The C ABI without Python: one docking-v0 env batch driven from plain C (HIP runtime for the device buffers only).
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_api_demo.c -Lquadsim_amd/csrc -lquadsim_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/quadsim_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_api_demo
 * Known answers: reset obs = (1.8, 0, ..., 0); with the hover action a = (-0.5)^4 the first
 * reward of an episode is -6 - 0.1 |a| = -6.1, and after the first step both drones fall at vz = -0.1962 (the stored
 * control is zero until the first command has been limited: the reference's one-step action delay). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "quadsim.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, qs_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(void)
{
    const int64_t n = 1000;
    QsConfig cfg;
    CK(qs_config_default(&cfg));
    cfg.kind = QS_KIND_DOCKING_V0; cfg.num_envs = n; cfg.auto_reset = 1; cfg.io_space = QS_IO_DEVICE;
    QsEnv *env = NULL;
    CK(qs_create(&cfg, &env));
    float *d_act, *d_obs, *d_rew; uint8_t *d_done;
    HK(hipMalloc((void **)&d_act, n * 4 * sizeof(float))); HK(hipMalloc((void **)&d_obs, n * 12 * sizeof(float)));
    HK(hipMalloc((void **)&d_rew, n * sizeof(float))); HK(hipMalloc((void **)&d_done, n));
    float *h_act = (float *)malloc(n * 4 * sizeof(float)), h_obs[12], h_rew, sc[13];
    for (int64_t i = 0; i < n * 4; ++i) h_act[i] = -0.5f;
    HK(hipMemcpy(d_act, h_act, n * 4 * sizeof(float), hipMemcpyHostToDevice));
    CK(qs_reset(env, NULL, d_obs));
    CK(qs_sync(env));
    HK(hipMemcpy(h_obs, d_obs, sizeof h_obs, hipMemcpyDeviceToHost));
    printf("reset obs[0] = %.6f\n", h_obs[0]);
    CK(qs_step(env, d_act, d_obs, d_rew, d_done, NULL, NULL));
    CK(qs_sync(env));
    HK(hipMemcpy(&h_rew, d_rew, sizeof h_rew, hipMemcpyDeviceToHost));
    float *d_sc; HK(hipMalloc((void **)&d_sc, n * 13 * sizeof(float)));
    CK(qs_get_state(env, d_sc, NULL, NULL, NULL, NULL, NULL));
    CK(qs_sync(env));
    HK(hipMemcpy(sc, d_sc, sizeof sc, hipMemcpyDeviceToHost));
    printf("first reward = %.6f\nchaser vz after step 1 = %.6f\n", h_rew, sc[5]);
    for (int k = 0; k < 200; ++k) CK(qs_step(env, d_act, d_obs, d_rew, d_done, NULL, NULL));
    uint64_t steps = 0;
    CK(qs_get_step_counter(env, &steps));
    printf("steps = %llu\n", (unsigned long long)steps);
    CK(qs_destroy(env));
    hipFree(d_act); hipFree(d_obs); hipFree(d_rew); hipFree(d_done); hipFree(d_sc); free(h_act);
    return 0;
}
