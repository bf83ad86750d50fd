/* CPU oracle (plain C) for the L-inf index/sign/clamp arithmetic of the PGD / APGD loops.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ and __graft_entry__.smoke()
 * as the bit-exact checker of the HIP kernels in robustvlm_amd/csrc/attack_kernels.hip.  Never
 * linked into or called from the product path.
 *
 * Restates (one IEEE-754 fp32 rounding per operation; build with -ffp-contract=off):
 *   ref_pgd_linf_update   <- train/pgd_train.py:38-59, vlm_eval/attacks/utils.py:10,21
 *   ref_apgd_linf_step    <- train/apgd_train.py:205-229 == autoattack/autopgd_base.py:328-341
 *   ref_apgd_controller   <- train/apgd_train.py:301-305, 320-355 == autopgd_base.py:388-446
 *   ref_apgd_select       <- the index assignments of the same lines, as one pass over the images
 * Pinned against tests/golden/pgd_linf_elementwise_*.npz and apgd_train_smallnet_*.npz
 * (outputs of the reference itself), see tests/test_oracle_c.py.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline float sgn(float a) { return (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f); } /* NaN,+-0 -> 0 */
static inline float clampf(float v, float lo, float hi) { v = v < lo ? lo : v; return v > hi ? hi : v; }

/* mode_max != 0: delta += step*v (mode='max'), else delta -= step*v (mode='min'). */
void ref_pgd_linf_update(const float* x, const float* g, float* delta, float* vel, size_t n,
                         float eps, float step, float mom, int mode_max) {
    for (size_t i = 0; i < n; ++i) {
        float gi = g[i];
        if (gi != gi) gi = 0.0f;                       /* pgd_train.py:40-42 */
        float s = sgn(gi);                             /* utils.py:21 */
        float mv = mom * vel[i];
        float v = sgn(mv + s);                         /* :46-47 */
        float sv = step * v;
        float d = mode_max ? (delta[i] + sv) : (delta[i] - sv);   /* :49-52 */
        d = clampf(d, -eps, eps);                      /* :56 */
        float xa = x[i] + d;
        xa = clampf(xa, 0.0f, 1.0f);
        d = xa - x[i];                                 /* :57-59 */
        delta[i] = d;
        vel[i] = v;
    }
}

/* x_adv_old is overwritten with the incoming x_adv (apgd_train.py:206-207). */
void ref_apgd_linf_step(const float* x, float* x_adv, float* x_adv_old, const float* grad,
                        const float* step /*[B]*/, float a, float eps, size_t n_per, int B) {
    const float one_minus_a = (float)(1.0 - (double)a);
    for (int b = 0; b < B; ++b) {
        const float st = step[b];
        for (size_t j = 0; j < n_per; ++j) {
            size_t i = (size_t)b * n_per + j;
            float xa = x_adv[i], xo = x_adv_old[i], xc = x[i];
            float grad2 = xa - xo;
            float lo = xc - eps, hi = xc + eps;
            float sg = st * sgn(grad[i]);
            float z = xa + sg;
            z = clampf(fminf(fmaxf(z, lo), hi), 0.0f, 1.0f);
            float t1 = (z - xa) * a;
            float t2 = grad2 * one_minus_a;
            float u = (xa + t1) + t2;
            u = clampf(fminf(fmaxf(u, lo), hi), 0.0f, 1.0f);
            x_adv_old[i] = xa;
            x_adv[i] = u;
        }
    }
}

/* Per-sample bookkeeping for iteration i.  k = current checkpoint length, do_check = (counter3==k)
 * (both are data-independent and tracked by the caller).  Outputs the three per-sample flags that
 * ref_apgd_select consumes. */
void ref_apgd_controller(int i, int B, int n_iter, int k, int do_check, const float* loss_i,
                         const uint8_t* pred, float* loss_steps /*[n_iter,B]*/, float* loss_best,
                         float* loss_best_last_check, float* reduced_last_check, float* step,
                         uint8_t* acc, uint8_t* f_notpred, uint8_t* f_improved, uint8_t* f_reduced) {
    for (int b = 0; b < B; ++b) {
        float y1 = loss_i[b];
        acc[b] = acc[b] < pred[b] ? acc[b] : pred[b];
        f_notpred[b] = pred[b] ? 0 : 1;
        loss_steps[(size_t)i * B + b] = y1;
        int imp = y1 > loss_best[b];
        if (imp) loss_best[b] = y1;
        f_improved[b] = (uint8_t)imp;
        int red = 0;
        if (do_check) {
            float t = 0.0f;
            for (int c = 0; c < k; ++c) {
                int r0 = i - c, r1 = i - c - 1;
                if (r0 < 0) r0 += n_iter;          /* python negative-index wrap, :117-122 */
                if (r1 < 0) r1 += n_iter;
                t += (loss_steps[(size_t)r0 * B + b] > loss_steps[(size_t)r1 * B + b]) ? 1.0f : 0.0f;
            }
            float thr = (float)((double)k * 0.75);
            float osc = (t <= thr) ? 1.0f : 0.0f;
            float noimp = (1.0f - reduced_last_check[b]) *
                          ((loss_best_last_check[b] >= loss_best[b]) ? 1.0f : 0.0f);
            float r = osc > noimp ? osc : noimp;
            reduced_last_check[b] = r;
            loss_best_last_check[b] = loss_best[b];
            if (r > 0.0f) { step[b] = step[b] / 2.0f; red = 1; }
        }
        f_reduced[b] = (uint8_t)red;
    }
}

void ref_apgd_select(float* x_adv, float* grad, float* x_best, float* grad_best, float* x_best_adv,
                     const uint8_t* f_notpred, const uint8_t* f_improved, const uint8_t* f_reduced,
                     size_t n_per, int B) {
    for (int b = 0; b < B; ++b)
        for (size_t j = 0; j < n_per; ++j) {
            size_t i = (size_t)b * n_per + j;
            if (f_notpred[b]) x_best_adv[i] = x_adv[i];
            if (f_improved[b]) { x_best[i] = x_adv[i]; grad_best[i] = grad[i]; }
            if (f_reduced[b]) { x_adv[i] = x_best[i]; grad[i] = grad_best[i]; }
        }
}
