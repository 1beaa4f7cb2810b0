// Reward / MoE-gating heads (moe_reward.py:226-285): one 256-thread workgroup per sample.
// Inputs are the two post-final-norm hidden rows of the sample and the outputs of the gating MLPs' last
// hidden layer; everything here is a handful of <=28-wide matvecs, softmaxes and weighted sums, so the
// kernel is latency-bound by construction.  All intermediate roundings follow the reference's bf16 ops;
// aspect_scores and score are fp32 (moe_reward.py:262,277).
#include "mjv_common.h"

namespace {

struct HeadsArgs {
  mjv_heads_desc d;
};

constexpr int MAX_OBJ = 64, MAX_ASP = 16;

// dot(row of bf16 weights, bf16 vector) with fp32 accumulation, one wave per output
MJV_DEV float wave_dot(const u16* __restrict__ w, const u16* __restrict__ v, int n, int lane) {
  float acc = 0.f;
  for (int c = lane * 8; c < n; c += 512) {
    float a[8], b[8];
    unpack8(*(const u32x4*)(w + c), a);
    unpack8(*(const u32x4*)(v + c), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] * b[j];
  }
  return wave_sum(acc);
}

__global__ __launch_bounds__(256) void reward_heads_kernel(HeadsArgs args) {
  const mjv_heads_desc& p = args.d;
  __shared__ float s_rew0[MAX_OBJ], s_rew[MAX_OBJ], s_crit[MAX_OBJ], s_w[MAX_OBJ], s_asp_logit[MAX_ASP], s_asp[MAX_ASP],
      s_ascore[MAX_ASP];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u16* h_r = p.hr + (long)b * p.ldh;
  const u16* g_a = p.ga + (long)b * p.ldg;
  const u16* g_c = p.gc + (long)b * p.ldg;

  // rewards = regression_layer(h_r)            (moe_reward.py:239)
  for (int k = wave; k < p.n_obj; k += 4) {
    const float v = wave_dot(p.w_reg + (long)k * p.hidden, h_r, p.hidden, lane);
    if (lane == 0) s_rew0[k] = rbf(v);
  }
  // last gating layers (Linear with bias)      (moe_reward.py:29-32,38-42)
  for (int k = wave; k < p.n_asp; k += 4) {
    const float v = wave_dot(p.wa + (long)k * p.gate_hidden, g_a, p.gate_hidden, lane);
    if (lane == 0) s_asp_logit[k] = rbf(v + bf2f(p.ba[k]));
  }
  for (int k = wave; k < p.n_obj; k += 4) {
    const float v = wave_dot(p.wc + (long)k * p.gate_hidden, g_c, p.gate_hidden, lane);
    if (lane == 0) s_crit[k] = rbf(v + bf2f(p.bc[k]));
  }
  __syncthreads();
  // rewards = rewards @ reward_transform_matrix (moe_reward.py:240)
  for (int j = tid; j < p.n_obj; j += 256) {
    float acc = 0.f;
    for (int k = 0; k < p.n_obj; ++k) acc += s_rew0[k] * bf2f(p.w_transform[(long)k * p.n_obj + j]);
    s_rew[j] = rbf(acc);
  }
  __syncthreads();
  if (tid == 0) {
    const float T = p.temperature;
    // aspect gating: softmax(x / T, dim=1) * logit_scale[0]     (moe_reward.py:34-35)
    {
      float z[MAX_ASP];
      float mx = -INFINITY;
      for (int i = 0; i < p.n_asp; ++i) { z[i] = rbf(s_asp_logit[i] / T); mx = fmaxf(mx, z[i]); }
      float sum = 0.f;
      for (int i = 0; i < p.n_asp; ++i) { z[i] = expf(z[i] - mx); sum += z[i]; }
      const float ls = bf2f(p.ls_a[0]);
      for (int i = 0; i < p.n_asp; ++i) s_asp[i] = rbf(rbf(z[i] / sum) * ls);
    }
    // per-aspect softmax over its criteria, weighted sums            (moe_reward.py:253-276)
    const float lsc = bf2f(p.ls_c[0]);
    float last = 0.f;
    for (int a = 0; a < p.n_asp; ++a) {
      const int o0 = p.group_offsets[a], o1 = p.group_offsets[a + 1];
      float mx = -INFINITY;
      for (int o = o0; o < o1; ++o) { s_w[o] = rbf(s_crit[p.group_index[o]] / T); mx = fmaxf(mx, s_w[o]); }
      float sum = 0.f;
      for (int o = o0; o < o1; ++o) { s_w[o] = expf(s_w[o] - mx); sum += s_w[o]; }
      float acc = 0.f;
      for (int o = o0; o < o1; ++o) {
        s_w[o] = rbf(rbf(s_w[o] / sum) * lsc);
        acc += rbf(s_rew[p.group_index[o]] * s_w[o]);
      }
      last = rbf(acc);
      s_ascore[a] = last;
    }
    float score = 0.f;
    for (int a = 0; a < p.n_asp; ++a) score += s_ascore[a] * s_asp[a];
    p.score[b] = score;
    p.weighted_last[b] = f2bf(last);
    if (p.packed34) p.packed34[(long)b * (1 + p.n_asp + p.n_obj)] = score;
  }
  __syncthreads();
  for (int j = tid; j < p.n_obj; j += 256) {
    p.rewards[(long)b * p.n_obj + j] = f2bf(s_rew[j]);
    p.criteria_gating[(long)b * p.n_obj + j] = f2bf(s_crit[j]);
    p.aspect_weights[(long)b * p.n_obj + j] = f2bf(s_w[j]);
    if (p.packed34) p.packed34[(long)b * (1 + p.n_asp + p.n_obj) + 1 + p.n_asp + j] = s_rew[j];
  }
  for (int a = tid; a < p.n_asp; a += 256) {
    p.aspect_gating[(long)b * p.n_asp + a] = f2bf(s_asp[a]);
    p.aspect_scores[(long)b * p.n_asp + a] = s_ascore[a];
    if (p.packed34) p.packed34[(long)b * (1 + p.n_asp + p.n_obj) + 1 + a] = s_ascore[a];
  }
}

}  // namespace

extern "C" int mjv_reward_heads_bf16(const mjv_heads_desc* d, void* stream) {
  MJV_REQUIRE(d && d->hr && d->hg && d->ga && d->gc && d->w_reg && d->w_transform && d->wa && d->ba && d->wc && d->bc && d->ls_a &&
                  d->ls_c && d->group_offsets && d->group_index, "heads: null input pointer");
  MJV_REQUIRE(d->rewards && d->criteria_gating && d->aspect_gating && d->aspect_weights && d->weighted_last &&
                  d->aspect_scores && d->score, "heads: null output pointer");
  MJV_REQUIRE(d->batch > 0 && d->n_obj > 0 && d->n_obj <= MAX_OBJ && d->n_asp > 0 && d->n_asp <= MAX_ASP,
              "heads: unsupported sizes n_obj=%d n_asp=%d", d->n_obj, d->n_asp);
  MJV_REQUIRE(d->hidden % 8 == 0 && d->gate_hidden % 8 == 0 && d->ldh % 8 == 0 && d->ldg % 8 == 0, "heads: alignment");
  MJV_REQUIRE(d->temperature > 0.f, "heads: temperature must be positive");
  HeadsArgs a;
  a.d = *d;
  hipStream_t s = (hipStream_t)stream;
  MjvProfScope ps("reward_heads", s, 0, 0);
  hipLaunchKernelGGL(reward_heads_kernel, dim3(d->batch), dim3(256), 0, s, a);
  return mjv_check_launch("reward_heads");
}
