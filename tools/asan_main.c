#include <stdio.h>
#include <stdlib.h>
#include "nbody_oracle.h"
#include "nbody.h"
int main(void) {
    for (int n = 0; n <= 1031; n += (n < 40 ? 1 : 97)) {
        ofloat4 *X = malloc(sizeof(ofloat4) * (n ? n : 1)), *V = calloc(n ? n : 1, sizeof(ofloat4)), *A = calloc(n ? n : 1, sizeof(ofloat4));
        nbody_fill_seeded((nbody_float4*)X, n, n % 2, 7 + n);
        oracle_step_inplace(X, A, V, n, 0.1f, 0.002f);
        oracle_step_jacobi(X, A, V, n, 0.1f, 0.002f);
        oracle_step_jacobi_f64acc(X, A, V, n, 0.1f, 0.002f);
        if (n > 3) { ofloat4* out = malloc(sizeof(ofloat4) * (n - 2)); oracle_accel_range(X, out, 1, n - 1, 2, n, 0.002f, 0); oracle_accel_range(X, out, 1, n - 1, 0, n - 3, 0.002f, 1); free(out); }
        float* V3 = calloc(3 * (n ? n : 1), sizeof(float)); oracle_step_legacy(X, V3, n); free(V3);
        int b = oracle_verify_still_bodies(V, X, n) + oracle_verify_equality4(V, X, n) + nbody_verify_still_bodies((nbody_float4*)V, (nbody_float4*)X, n) + nbody_verify_equality4((nbody_float4*)V, (nbody_float4*)X, n);
        odouble4 *Xd = calloc(n ? n : 1, sizeof(odouble4)), *Vd = calloc(n ? n : 1, sizeof(odouble4)), *Ad = calloc(n ? n : 1, sizeof(odouble4));
        for (int i = 0; i < n; ++i) { Xd[i].x = X[i].x; Xd[i].y = X[i].y; Xd[i].z = X[i].z; Xd[i].w = X[i].w; }
        oracle_step_jacobi_f64(Xd, Ad, Vd, n, 0.01, 0.002);
        nbody_fill_with_random4((nbody_float4*)X, n); nbody_fill_with_zeroes4((nbody_float4*)X, n); oracle_fill_with_random4(X, n); oracle_fill_with_zeroes4(X, n);
        (void)b; free(X); free(V); free(A); free(Xd); free(Vd); free(Ad);
    }
    /* the shard plan (pure host logic of the multi-GPU step): every world size, every rank, the three schedules */
    for (int world = 1; world <= NBODY_MAX_RANKS; ++world)
        for (int rank = 0; rank < world; ++rank)
            for (int sched = 0; sched < 3; ++sched) {
                nbody_shard_plan_t p;
                if (nbody_shard_plan(rank, world, 1000 + 37 * world, sched, &p) != NBODY_OK) { puts("plan failed"); return 1; }
                if (p.n_sends > NBODY_MAX_RANKS || p.n_recvs > NBODY_MAX_RANKS || p.n_launches > 2) { puts("plan overflow"); return 1; }
            }
    { nbody_shard_plan_t p; if (nbody_shard_plan(0, 0, 10, 2, &p) == NBODY_OK || nbody_shard_plan(3, 2, 10, 2, &p) == NBODY_OK) return 1; }
    { odouble4 Xd[5] = {{0,0,0,1},{1,0,0,1},{0,1,0,1},{0,0,1,1},{1,1,1,1}}, out[3]; oracle_accel_range_f64(Xd, out, 1, 4, 0, 5, 0.002); }
    { nbody_float3 v3[4], w3[4]; nbody_fill_with_zeroes3(v3, 4); nbody_fill_with_zeroes3(w3, 4); (void)nbody_verify_equality3(v3, w3, 4); (void)nbody_random_float(0.f, 1.f); }
    puts("asan/ubsan: clean");
    return 0;
}
