/* ORACLE (test infrastructure, never shipped, never linked into libwtk_hip.so).
 *
 * Plain-C restatement of the reference's ResMLP inference, layer by layer, WITHOUT BatchNorm folding:
 *   RMLP.forward            wtracker/neural/mlp.py:184-188   x = input(x); x = x + block(x) ...; output(x)
 *   MlpBlock / MLPLayer     wtracker/neural/mlp.py:121-126, 67-71   Linear -> BatchNorm1d(eval) -> ReLU
 * Pinned by tests/golden/resmlp_{100,200}ms.npz (outputs of the real reference), see
 * tests/test_oracle_resmlp.py.  Also used as the scalar CPU baseline of the predictor.
 *
 * Layout of `params` for each MLPLayer, in execution order (input, block0.l0.., ..., output):
 *   W[out][in], b[out], then (if has_bn) gamma[out], beta[out], running_mean[out], running_var[out].
 */
#include <math.h>
#include <stddef.h>

#define MAX_DIM 256

typedef struct {
    int in_dim, out_dim, has_bn;
} layer_desc;

static const float *apply_layer(const layer_desc *d, const float *p, const float *x, float *y) {
    const float *W = p, *b = p + (size_t)d->out_dim * d->in_dim;
    const float *q = b + d->out_dim;
    for (int o = 0; o < d->out_dim; ++o) {
        float s = 0.0f;
        for (int k = 0; k < d->in_dim; ++k) s += W[(size_t)o * d->in_dim + k] * x[k];
        s += b[o];
        if (d->has_bn) {
            const float g = q[o], beta = q[d->out_dim + o], mu = q[2 * d->out_dim + o], var = q[3 * d->out_dim + o];
            s = (s - mu) / sqrtf(var + 1e-5f) * g + beta; /* eval-mode BatchNorm1d */
            s = s > 0.0f ? s : 0.0f;                      /* ReLU */
        }
        y[o] = s;
    }
    return q + (d->has_bn ? 4 * d->out_dim : 0);
}

/* x [batch][in], y [batch][out]; layers: 1 + n_blocks*per_block + 1 descriptors. Returns 0 on success. */
int resmlp_forward(const layer_desc *layers, int n_blocks, int per_block, const float *params, const float *x, int batch,
                   float *y) {
    float h[MAX_DIM], t0[MAX_DIM], t1[MAX_DIM];
    const int n_layers = 2 + n_blocks * per_block;
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].in_dim > MAX_DIM || layers[i].out_dim > MAX_DIM) return 1;
    const int in_dim = layers[0].in_dim, out_dim = layers[n_layers - 1].out_dim, hid = layers[0].out_dim;
    for (int n = 0; n < batch; ++n) {
        const float *p = apply_layer(&layers[0], params, x + (size_t)n * in_dim, h);
        int li = 1;
        for (int b = 0; b < n_blocks; ++b) {
            const float *src = h;
            float *dst = t0;
            for (int l = 0; l < per_block; ++l, ++li) {
                p = apply_layer(&layers[li], p, src, dst);
                src = dst;
                dst = (dst == t0) ? t1 : t0;
            }
            for (int k = 0; k < hid; ++k) h[k] += src[k];
        }
        apply_layer(&layers[li], p, h, y + (size_t)n * out_dim);
    }
    return 0;
}
