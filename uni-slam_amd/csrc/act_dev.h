// act_dev.h -- the decoders' output activations (us_mlp_desc.out_act) and their derivatives, shared by the decoder kernels (mlp.hip) and the
// compositing kernels (render.hip): with US_MLP_OUT_PREACT / US_RENDER_ACT the activation is evaluated by the consumer of `raw`, value for
// value what the decoder would have written.
#ifndef US_ACT_DEV_H
#define US_ACT_DEV_H
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == US_ACT_TANH) return tanhf(v);
    if (act == US_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == US_ACT_TANH) return 1.0f - y * y;
    if (act == US_ACT_SIGMOID) return y * (1.0f - y);
    return 1.0f;
}
#endif
