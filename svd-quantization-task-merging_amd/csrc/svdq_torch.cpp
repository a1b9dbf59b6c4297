// libsvdq_torch.so -- torch.ops.svdq.*: the C ABI of libsvdq_hip.so (include/svdq.h) registered as PyTorch custom
// operators with TORCH_LIBRARY (BASELINE.json north_star: "exposed as PyTorch-ROCm custom ops").
//
// Host code only: every operator checks its tensors, allocates outputs with the caching allocator, takes the current
// HIP stream of the tensors' device and enqueues the hand-written kernels through the C ABI.  There is no ATen
// arithmetic behind any of them and no CPU kernel is registered: CPU tensors are refused by the dispatcher.
//
// Schemas (the reference callables each one stands for are cited at the C entry point it calls, include/svdq.h):
//   rtvq_quantize(Tensor x, int bits, int stages) -> (Tensor codes, Tensor scale, Tensor zero_point, Tensor rnorm)
//   rtvq_dequantize(Tensor codes, Tensor scale, Tensor zero_point) -> Tensor
//   mask_combine(Tensor[] masks, str strategy) -> Tensor
//   mask_select(Tensor x, Tensor mask, bool invert) -> Tensor
//   compress(Tensor[] deltas, int n_tasks, float energy, int max_rank, bool center, bool fp16, int bits, int stages)
//       -> (Tensor small, Tensor basis, Tensor mean)          packed buffers: svdq_plan_small_layout / _basis_layout
//   compress_masked(Tensor[] deltas, Tensor[] masks, int n_tasks, <settings>) -> (small, basis, mean, Tensor rows)
//       masks[p] = the combined mask of parameter p; mask walk (svdq_compress_masked)
//   compress_gather(...same...)                                the same through int32 index lists (svdq_compress_gather)
//   compress_from_base(Tensor[] finetuned, Tensor[] base, int n_tasks, <settings>) -> (small, basis, mean)
//   mask_combine_indices(Tensor[] masks, int n_masks, str strategy) -> (Tensor[] combined, Tensor[] indices, Tensor counts)
//   reconstruct(Tensor U_high, Tensor U_low, Tensor coef, Tensor? mean, float scale) -> Tensor
//   recon_error(Tensor U_high, Tensor U_low, Tensor coef, Tensor? mean, Tensor orig) -> Tensor      (float64 [6])
//   merge(Tensor small, Tensor basis, Tensor mean, int[] rows, int n_tasks, <settings>, Tensor weights, Tensor[] base)
//       -> Tensor[]                                            svdq_merge on the buffers `compress` returned
//   merge_masked(Tensor small, Tensor basis, Tensor mean, Tensor[] masks, int n_tasks, <settings>, Tensor weights,
//       Tensor[] base) -> Tensor[]                             svdq_merge_masked: full-size tensors, mask scatter fused
//   diagnostics(Tensor[] deltas, Tensor[] masks, Tensor small, Tensor basis, Tensor mean, int n_tasks, <settings>,
//       bool add_mean) -> Tensor                               float64 [P, n_tasks, 6]; masks empty = unmasked
//   ingest(Tensor base, Tensor[] finetuned) -> Tensor[]
//   task_gram(Tensor[] deltas, int n_tasks) -> Tensor
//   plan_cache_size() -> int                                  plans kept by compress (for tests)
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cstdint>
#include <list>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "svdq.h"

namespace {

void check(int rc, const char *what) {
    if (rc == SVDQ_OK) return;
    const char *msg = svdq_last_error();
    // bad arguments are ValueError in the reference (SURVEY 8b "Errors"); TORCH_CHECK_VALUE raises exactly that
    TORCH_CHECK_VALUE(rc != SVDQ_EINVAL, what, ": ", msg ? msg : "invalid argument");
    TORCH_CHECK(false, what, " failed (", rc, "): ", msg ? msg : "");
}

void *stream_of(const c10::Device &dev) { return (void *)c10::hip::getCurrentHIPStream(dev.index()).stream(); }

void sync(const c10::Device &dev) { c10::hip::getCurrentHIPStream(dev.index()).synchronize(); }

// fp32, flat, contiguous, 16-byte aligned: the only input contract of the ABI
at::Tensor prep(const at::Tensor &t) {
    TORCH_CHECK(t.is_cuda(), "svdq operators take device tensors");
    at::Tensor v = t.detach();
    if (v.scalar_type() != at::kFloat) v = v.to(at::kFloat);
    v = v.contiguous().view({-1});
    if (reinterpret_cast<uintptr_t>(v.data_ptr()) & 15) v = v.clone();
    return v;
}

at::Tensor bytes_on(const c10::Device &dev, int64_t n) {
    return at::empty({n}, at::TensorOptions().dtype(at::kByte).device(dev));
}

at::Tensor floats_on(const c10::Device &dev, int64_t n) {
    return at::empty({n}, at::TensorOptions().dtype(at::kFloat).device(dev));
}

// device array of the tensors' base addresses (the ABI's pointer tables)
at::Tensor table_of(const std::vector<at::Tensor> &ts, const c10::Device &dev) {
    std::vector<int64_t> p(ts.size());
    for (size_t i = 0; i < ts.size(); ++i) p[i] = reinterpret_cast<int64_t>(ts[i].data_ptr());
    return at::tensor(p, at::TensorOptions().dtype(at::kLong)).to(dev);
}

at::Tensor mask_bytes(const at::Tensor &m) {
    TORCH_CHECK(m.is_cuda(), "svdq operators take device tensors");
    at::Tensor b = (m.scalar_type() == at::kBool || m.scalar_type() == at::kByte) ? m : m.to(at::kBool);
    b = b.contiguous().view({-1});
    return b.scalar_type() == at::kBool ? b.view(at::kByte) : b;
}

// ------------------------------------------------------------------------------------------ quantizer
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> rtvq_quantize(const at::Tensor &x, int64_t bits,
                                                                         int64_t stages) {
    const at::Tensor v = prep(x);
    const c10::Device dev = v.device();
    c10::DeviceGuard guard(dev);
    const int64_t n = v.numel();
    TORCH_CHECK_VALUE(n >= 1, "rtvq_quantize: empty tensor");
    TORCH_CHECK_VALUE(stages >= 1 && stages <= SVDQ_MAX_STAGES, "rtvq_quantize: stages out of range");
    const int64_t stride = (n + 3) / 4 * 4;
    at::Tensor codes = at::empty({stages, stride}, at::TensorOptions().dtype(at::kByte).device(dev));
    at::Tensor scale = floats_on(dev, stages), zp = floats_on(dev, stages), rnorm = floats_on(dev, stages);
    at::Tensor work = bytes_on(dev, svdq_rtvq_work_bytes(n));
    check(svdq_rtvq_quantize(v.data_ptr<float>(), n, (int32_t)bits, (int32_t)stages, codes.data_ptr<uint8_t>(), stride,
                             scale.data_ptr<float>(), zp.data_ptr<float>(), rnorm.data_ptr<float>(), work.data_ptr(),
                             stream_of(dev)),
          "svdq_rtvq_quantize");
    return {codes.narrow(1, 0, n).contiguous(), scale, zp, rnorm};
}

at::Tensor rtvq_dequantize(const at::Tensor &codes_in, const at::Tensor &scale_in, const at::Tensor &zp_in) {
    TORCH_CHECK(codes_in.is_cuda(), "svdq operators take device tensors");
    const c10::Device dev = codes_in.device();
    c10::DeviceGuard guard(dev);
    at::Tensor codes = codes_in.dim() == 1 ? codes_in.unsqueeze(0) : codes_in;
    TORCH_CHECK_VALUE(codes.dim() == 2 && codes.scalar_type() == at::kByte, "rtvq_dequantize: codes are uint8 [stages, n]");
    const int64_t stages = codes.size(0), n = codes.size(1);
    at::Tensor scale = scale_in.to(dev, at::kFloat).reshape({-1}).contiguous();
    at::Tensor zp = zp_in.to(dev, at::kFloat).reshape({-1}).contiguous();
    TORCH_CHECK_VALUE(scale.numel() == stages && zp.numel() == stages, "rtvq_dequantize: one scale / zero point per stage");
    const int64_t stride = (n + 3) / 4 * 4;
    if (stride != n || !codes.is_contiguous() || (reinterpret_cast<uintptr_t>(codes.data_ptr()) & 3)) {
        at::Tensor padded = at::zeros({stages, stride}, codes.options());
        padded.narrow(1, 0, n).copy_(codes);
        codes = padded;
    }
    at::Tensor out = floats_on(dev, stride);
    if (n > 0)
        check(svdq_rtvq_dequantize(codes.data_ptr<uint8_t>(), stride, n, (int32_t)stages, scale.data_ptr<float>(),
                                   zp.data_ptr<float>(), out.data_ptr<float>(), stream_of(dev)),
              "svdq_rtvq_dequantize");
    return out.narrow(0, 0, n);
}

// ------------------------------------------------------------------------------------------ masks
at::Tensor mask_combine(at::TensorList masks, c10::string_view strategy) {
    TORCH_CHECK_VALUE(!masks.empty(), "Empty mask list");
    int32_t code;
    if (strategy == "union") code = SVDQ_MASK_UNION;
    else if (strategy == "intersection") code = SVDQ_MASK_INTERSECTION;
    else if (strategy == "majority") code = SVDQ_MASK_MAJORITY;
    else TORCH_CHECK_VALUE(false, "Unknown mask strategy: ", std::string(strategy));
    for (const at::Tensor &m : masks) {
        TORCH_CHECK_VALUE(m.sizes() == masks[0].sizes(), "Shape mismatch: mask ", m.sizes(), " vs mask ", masks[0].sizes());
        // the kernel dereferences every mask from masks[0]'s device
        TORCH_CHECK_VALUE(m.device() == masks[0].device(), "mask_combine: all masks must live on one device (got ",
                          m.device(), " and ", masks[0].device(), ")");
    }
    const c10::Device dev = masks[0].device();
    c10::DeviceGuard guard(dev);
    std::vector<at::Tensor> flat;
    for (const at::Tensor &m : masks) flat.push_back(mask_bytes(m));
    const int64_t numel = flat[0].numel();
    at::Tensor out = bytes_on(dev, numel);
    if (numel > 0) {
        at::Tensor table = table_of(flat, dev);
        at::Tensor count = at::zeros({1}, at::TensorOptions().dtype(at::kLong).device(dev));
        at::Tensor work = bytes_on(dev, svdq_mask_work_bytes(numel));
        check(svdq_mask_combine(table.data_ptr(), (int32_t)flat.size(), numel, code, out.data_ptr<uint8_t>(),
                                count.data_ptr<int64_t>(), work.data_ptr(), stream_of(dev)),
              "svdq_mask_combine");
        // `flat`, `table`, `work` die here; the caching allocator releases them stream-ordered on this same stream
    }
    return out.view(at::kBool).view(masks[0].sizes());
}

at::Tensor mask_select(const at::Tensor &x, const at::Tensor &mask, bool invert) {
    TORCH_CHECK_VALUE(x.sizes() == mask.sizes(), "Shape mismatch: tensor ", x.sizes(), " vs mask ", mask.sizes());
    if (x.numel() == 0) return x.flatten();
    if (x.scalar_type() != at::kFloat) return mask_select(x.to(at::kFloat), mask, invert).to(x.scalar_type());
    const at::Tensor v = prep(x);
    const c10::Device dev = v.device();
    c10::DeviceGuard guard(dev);
    const int64_t numel = v.numel();
    at::Tensor mb = mask_bytes(mask.to(dev));
    at::Tensor dst = floats_on(dev, numel);
    at::Tensor count = at::zeros({1}, at::TensorOptions().dtype(at::kLong).device(dev));
    at::Tensor stab = table_of({v}, dev), dtab = table_of({dst}, dev);
    at::Tensor work = bytes_on(dev, svdq_mask_work_bytes(numel));
    check(svdq_mask_compact(stab.data_ptr(), dtab.data_ptr(), 1, mb.data_ptr<uint8_t>(), invert ? 1 : 0, numel,
                            count.data_ptr<int64_t>(), work.data_ptr(), stream_of(dev)),
          "svdq_mask_compact");
    const int64_t n = count.item<int64_t>();          // the output size of boolean indexing is data dependent
    return dst.narrow(0, 0, n).clone();
}

// ------------------------------------------------------------------------------------------ plans
struct Plan {
    svdq_plan *h = nullptr;
    svdq_sizes sizes{};
    at::Tensor workspace;
    c10::Device dev{c10::kCUDA, 0};
    Plan() = default;
    Plan(const Plan &) = delete;
    Plan &operator=(const Plan &) = delete;
    ~Plan() {
        if (h) svdq_plan_destroy(h);
    }
};

std::unique_ptr<Plan> make_plan(const std::vector<int64_t> &rows, int64_t n_tasks, const svdq_config &cfg,
                                const c10::Device &dev) {
    auto p = std::make_unique<Plan>();
    p->dev = dev;
    check(svdq_plan_create(&p->h, (int32_t)n_tasks, (int32_t)rows.size(), rows.data(), &cfg), "svdq_plan_create");
    check(svdq_plan_sizes(p->h, &p->sizes), "svdq_plan_sizes");
    p->workspace = bytes_on(dev, p->sizes.workspace_bytes);
    return p;
}

// vecs[p * n_tasks + t], checked against each other: one row count per parameter
std::vector<int64_t> rows_of(const std::vector<at::Tensor> &vecs, int64_t n_tasks) {
    const int64_t P = (int64_t)vecs.size() / n_tasks;
    std::vector<int64_t> rows(P);
    for (int64_t p = 0; p < P; ++p) {
        rows[p] = vecs[p * n_tasks].numel();
        for (int64_t t = 1; t < n_tasks; ++t)
            TORCH_CHECK_VALUE(vecs[p * n_tasks + t].numel() == rows[p], "parameter ", p, ": task tensors differ in size");
    }
    return rows;
}

std::vector<at::Tensor> prep_list(at::TensorList ts, int64_t n_tasks, const char *what) {
    TORCH_CHECK_VALUE(n_tasks >= 1 && !ts.empty() && (int64_t)ts.size() % n_tasks == 0, what,
                      ": deltas must hold n_tasks tensors per parameter (parameter-major)");
    std::vector<at::Tensor> v;
    v.reserve(ts.size());
    for (const at::Tensor &t : ts) {
        TORCH_CHECK_VALUE(t.device() == ts[0].device(), what, ": all tensors must live on one device");
        v.push_back(prep(t));
    }
    return v;
}

// Plans (device tables + workspace) are kept per (sizes, N, settings, device, stream): a repeated call with the same
// shapes creates nothing and does not synchronise -- its kernels are ordered behind the previous call's on the same
// stream, which is also what makes sharing the workspace safe.  Output buffers are fresh per call (the caller owns them).
struct PlanKey {
    std::vector<int64_t> rows;
    int64_t n_tasks, max_rank, bits, stages;
    double energy;
    bool center, fp16;
    int dev;
    void *stream;
    bool operator==(const PlanKey &o) const {
        return rows == o.rows && n_tasks == o.n_tasks && max_rank == o.max_rank && bits == o.bits && stages == o.stages &&
               energy == o.energy && center == o.center && fp16 == o.fp16 && dev == o.dev && stream == o.stream;
    }
};
constexpr size_t kPlanCacheMax = 8;
std::mutex g_cache_mu;
using PlanCache = std::list<std::pair<PlanKey, std::unique_ptr<Plan>>>;
// most recently used last; never destroyed: at process exit the HIP runtime may be gone before static destructors run
PlanCache &g_cache = *new PlanCache;

// take the plan of `key` out of the cache (or make it); give it back with release_plan once the launches are enqueued
std::unique_ptr<Plan> acquire_plan(const PlanKey &key, const c10::Device &dev) {
    std::unique_ptr<Plan> plan;
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
        if (it->first == key) {
            plan = std::move(it->second);
            g_cache.erase(it);
            break;
        }
    if (!plan) {
        svdq_config cfg{};
        cfg.energy_threshold = (float)key.energy;
        cfg.max_rank = key.max_rank > 0 ? (int32_t)key.max_rank : 0;
        cfg.center = key.center;
        cfg.fp16 = key.fp16;
        cfg.low_bits = (int32_t)key.bits;
        cfg.rtvq_stages = (int32_t)key.stages;
        plan = make_plan(key.rows, key.n_tasks, cfg, dev);
        while (g_cache.size() >= kPlanCacheMax) {
            sync(g_cache.front().second->dev);                        // its tables may still be in use
            g_cache.pop_front();
        }
    }
    return plan;
}

void release_plan(PlanKey key, std::unique_ptr<Plan> plan) { g_cache.emplace_back(std::move(key), std::move(plan)); }

struct Outputs {
    at::Tensor small, basis, mean;
};

// basis and mean in ONE allocation, the mean right behind the basis (pass-2 time depends on it: DESIGN.md section 5)
Outputs alloc_outputs(const Plan &plan, bool center, const c10::Device &dev) {
    const int64_t bb = plan.sizes.basis_bytes, gap = (bb + 255) / 256 * 256;
    const int64_t nm = center ? plan.sizes.mean_floats * 4 : 0;
    at::Tensor out = bytes_on(dev, gap + nm);
    Outputs o;
    o.basis = out.narrow(0, 0, bb);
    o.mean = center ? out.narrow(0, gap, nm).view(at::kFloat) : floats_on(dev, 0);
    o.small = at::zeros({plan.sizes.small_bytes}, at::TensorOptions().dtype(at::kByte).device(dev));
    return o;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> compress(at::TensorList deltas, int64_t n_tasks, double energy,
                                                        int64_t max_rank, bool center, bool fp16, int64_t bits,
                                                        int64_t stages) {
    std::vector<at::Tensor> vecs = prep_list(deltas, n_tasks, "compress");
    const c10::Device dev = vecs[0].device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    PlanKey key{rows_of(vecs, n_tasks), n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan = acquire_plan(key, dev);
    Outputs o = alloc_outputs(*plan, center, dev);
    at::Tensor table = table_of(vecs, dev);
    const int rc = svdq_compress(plan->h, table.data_ptr(), nullptr, plan->workspace.data_ptr(), o.small.data_ptr(),
                                 o.basis.data_ptr(), center ? o.mean.data_ptr<float>() : nullptr, stream);
    release_plan(std::move(key), std::move(plan));
    check(rc, "svdq_compress");
    // inputs and the table are only read by the kernels just enqueued; temporaries are released stream-ordered
    return {o.small, o.basis, o.mean};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> compress_from_base(at::TensorList finetuned, at::TensorList base,
                                                                  int64_t n_tasks, double energy, int64_t max_rank,
                                                                  bool center, bool fp16, int64_t bits, int64_t stages) {
    std::vector<at::Tensor> vecs = prep_list(finetuned, n_tasks, "compress_from_base");
    const c10::Device dev = vecs[0].device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    std::vector<int64_t> rows = rows_of(vecs, n_tasks);
    TORCH_CHECK_VALUE(base.size() == rows.size(), "compress_from_base: one base tensor per parameter");
    std::vector<at::Tensor> bs;
    for (size_t p = 0; p < rows.size(); ++p) {
        TORCH_CHECK_VALUE(base[p].device() == dev, "compress_from_base: all tensors must live on one device");
        TORCH_CHECK_VALUE(base[p].numel() == rows[p], "parameter ", p, ": base and fine-tuned tensors differ in size");
        bs.push_back(prep(base[p]));
    }
    PlanKey key{rows, n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan = acquire_plan(key, dev);
    Outputs o = alloc_outputs(*plan, center, dev);
    at::Tensor table = table_of(vecs, dev), btab = table_of(bs, dev);
    const int rc = svdq_compress_from_base(plan->h, table.data_ptr(), btab.data_ptr(), nullptr, plan->workspace.data_ptr(),
                                           o.small.data_ptr(), o.basis.data_ptr(),
                                           center ? o.mean.data_ptr<float>() : nullptr, stream);
    release_plan(std::move(key), std::move(plan));
    check(rc, "svdq_compress_from_base");
    return {o.small, o.basis, o.mean};
}

// masked parameters: masks[p] = the combined mask of parameter p (bool / uint8, the parameter's shape).
// walk = true: svdq_maskset_count_scan + _unit_starts + svdq_compress_masked (no index lists; above 16 tasks on the one-wave kernels);
// walk = false: svdq_maskset_indices + svdq_compress_gather.  rows (int64 [P], device) = mask.sum() per parameter.
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> compress_with_masks(at::TensorList deltas, at::TensorList masks,
                                                                               int64_t n_tasks, double energy,
                                                                               int64_t max_rank, bool center, bool fp16,
                                                                               int64_t bits, int64_t stages, bool walk,
                                                                               const char *what) {
    std::vector<at::Tensor> vecs = prep_list(deltas, n_tasks, what);
    const c10::Device dev = vecs[0].device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    std::vector<int64_t> rows = rows_of(vecs, n_tasks);
    const int64_t P = (int64_t)rows.size();
    TORCH_CHECK_VALUE((int64_t)masks.size() == P, what, ": one combined mask per parameter");
    std::vector<at::Tensor> mb;
    for (int64_t p = 0; p < P; ++p) {
        TORCH_CHECK_VALUE(masks[p].device() == dev, what, ": all tensors must live on one device");
        TORCH_CHECK_VALUE(masks[p].numel() == rows[p], "Shape mismatch: tensor vs mask for parameter ", p);
        mb.push_back(mask_bytes(masks[p]));
    }
    svdq_maskset *ms = nullptr;
    check(svdq_maskset_create(&ms, (int32_t)P, rows.data()), "svdq_maskset_create");
    struct Guard {
        svdq_maskset *m;
        c10::Device d;
        ~Guard() {
            sync(d);      // its device tables are read by the launches below
            svdq_maskset_destroy(m);
        }
    } ms_guard{ms, dev};
    at::Tensor work = bytes_on(dev, svdq_maskset_work_bytes(ms));
    at::Tensor ct = at::zeros({P}, at::TensorOptions().dtype(at::kLong).device(dev));
    at::Tensor mtab = table_of(mb, dev);
    PlanKey key{rows, n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan = acquire_plan(key, dev);
    Outputs o = alloc_outputs(*plan, center, dev);
    at::Tensor table = table_of(vecs, dev);
    int rc;
    if (walk) {
        at::Tensor us = at::empty({plan->sizes.n_units}, at::TensorOptions().dtype(at::kLong).device(dev));
        rc = svdq_maskset_count_scan(ms, mtab.data_ptr(), ct.data_ptr<int64_t>(), nullptr, work.data_ptr(), stream);
        if (rc == SVDQ_OK)
            rc = svdq_maskset_unit_starts(ms, plan->h, mtab.data_ptr(), nullptr, ct.data_ptr<int64_t>(), work.data_ptr(),
                                          us.data_ptr<int64_t>(), stream);
        if (rc == SVDQ_OK)
            rc = svdq_compress_masked(plan->h, table.data_ptr(), mtab.data_ptr(), us.data_ptr<int64_t>(),
                                      ct.data_ptr<int64_t>(), plan->workspace.data_ptr(), o.small.data_ptr(),
                                      o.basis.data_ptr(), center ? o.mean.data_ptr<float>() : nullptr, stream);
    } else {
        std::vector<at::Tensor> idx;
        for (int64_t p = 0; p < P; ++p) idx.push_back(at::empty({rows[p]}, at::TensorOptions().dtype(at::kInt).device(dev)));
        at::Tensor itab = table_of(idx, dev);
        rc = svdq_maskset_indices(ms, mtab.data_ptr(), itab.data_ptr(), nullptr, ct.data_ptr<int64_t>(), nullptr,
                                  work.data_ptr(), stream);
        if (rc == SVDQ_OK)
            rc = svdq_compress_gather(plan->h, table.data_ptr(), itab.data_ptr(), ct.data_ptr<int64_t>(),
                                      plan->workspace.data_ptr(), o.small.data_ptr(), o.basis.data_ptr(),
                                      center ? o.mean.data_ptr<float>() : nullptr, stream);
        sync(dev);      // the index lists die with this scope
    }
    release_plan(std::move(key), std::move(plan));
    check(rc, what);
    return {o.small, o.basis, o.mean, ct};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> compress_masked(at::TensorList deltas, at::TensorList masks,
                                                                           int64_t n_tasks, double energy,
                                                                           int64_t max_rank, bool center, bool fp16,
                                                                           int64_t bits, int64_t stages) {
    return compress_with_masks(deltas, masks, n_tasks, energy, max_rank, center, fp16, bits, stages, true,
                               "compress_masked");
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> compress_gather(at::TensorList deltas, at::TensorList masks,
                                                                           int64_t n_tasks, double energy,
                                                                           int64_t max_rank, bool center, bool fp16,
                                                                           int64_t bits, int64_t stages) {
    return compress_with_masks(deltas, masks, n_tasks, energy, max_rank, center, fp16, bits, stages, false,
                               "compress_gather");
}

// the packed buffers a consumer operator is handed must be the ones a plan with these settings wrote: on the
// operator's device, contiguous, of the plan's sizes -- a short or foreign `mean` would otherwise be read out of bounds by
// the streaming kernels instead of raising here
static void check_artifacts(const char *what, const at::Tensor &small, const at::Tensor &basis, const at::Tensor &mean,
                            const svdq_sizes &sz, bool center, const c10::Device &dev) {
    TORCH_CHECK_VALUE(small.device() == dev && basis.device() == dev, what, ": all tensors must live on one device");
    TORCH_CHECK_VALUE(small.is_contiguous() && basis.is_contiguous(), what, ": small / basis must be contiguous");
    TORCH_CHECK_VALUE((int64_t)small.nbytes() == sz.small_bytes && (int64_t)basis.nbytes() >= sz.basis_bytes, what,
                      ": small / basis are not the buffers of a plan with these rows and settings");
    if (center) {
        TORCH_CHECK_VALUE(mean.defined() && mean.device() == dev && mean.scalar_type() == at::kFloat && mean.is_contiguous() &&
                              mean.numel() >= sz.mean_floats,
                          what, ": mean must be the plan's float32 mean buffer (", sz.mean_floats,
                          " floats, contiguous, on the same device) when center is set");
    }
}

// svdq_merge on the buffers `compress` returned for the same rows / settings: merged delta of every parameter
// (weights float32 [n_tasks] or [n_sets, n_tasks] with shares = uniform over sets when more than one), + base when given.
// The weights are used AS GIVEN (the kernel adds w_t c_t over the tasks): like merge.py:123-124 the caller renormalises
// them over the tasks that are present; a negative entry marks a task that is not in the set.
std::vector<at::Tensor> merge(const at::Tensor &small, const at::Tensor &basis, const at::Tensor &mean,
                              at::IntArrayRef rows_in, int64_t n_tasks, double energy, int64_t max_rank, bool center,
                              bool fp16, int64_t bits, int64_t stages, const at::Tensor &weights, at::TensorList base) {
    TORCH_CHECK(small.is_cuda() && basis.is_cuda(), "svdq operators take device tensors");
    const c10::Device dev = small.device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    std::vector<int64_t> rows(rows_in.begin(), rows_in.end());
    const int64_t P = (int64_t)rows.size();
    TORCH_CHECK_VALUE(P >= 1, "Empty delta list");
    TORCH_CHECK_VALUE(base.empty() || (int64_t)base.size() == P, "merge: one base tensor per parameter, or none");
    at::Tensor w = weights.to(dev, at::kFloat).contiguous();
    TORCH_CHECK_VALUE((w.dim() == 1 || w.dim() == 2) && w.size(-1) == n_tasks, "merge: weights are [n_tasks] or [n_sets, n_tasks]");
    const int64_t n_sets = w.dim() == 2 ? w.size(0) : 1;
    PlanKey key{rows, n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan = acquire_plan(key, dev);
    check_artifacts("merge", small, basis, mean, plan->sizes, center, dev);
    std::vector<at::Tensor> outs, bs;
    for (int64_t p = 0; p < P; ++p) outs.push_back(floats_on(dev, rows[p]));
    for (size_t p = 0; p < base.size(); ++p) {
        TORCH_CHECK_VALUE(base[p].numel() == rows[p] && base[p].device() == dev, "merge: base tensor ", p, " does not match");
        bs.push_back(prep(base[p]));
    }
    at::Tensor otab = table_of(outs, dev), btab = bs.empty() ? at::Tensor() : table_of(bs, dev);
    at::Tensor share;
    if (n_sets > 1) share = at::full({n_sets}, 1.0 / (double)n_sets, at::TensorOptions().dtype(at::kFloat).device(dev));
    at::Tensor work = bytes_on(dev, svdq_merge_work_bytes(plan->h, (int32_t)n_sets));
    const int64_t rows_off = [&] {
        svdq_small_layout L{};
        svdq_plan_small_layout(plan->h, &L);
        return L.rows_off;
    }();
    const int rc = svdq_merge(plan->h, reinterpret_cast<const int64_t *>(small.data_ptr<uint8_t>() + rows_off),
                              small.data_ptr(), basis.data_ptr(), (center && mean.numel() > 0) ? mean.data_ptr<float>() : nullptr,
                              w.data_ptr<float>(), nullptr, (int32_t)n_sets, 0,
                              n_sets > 1 ? share.data_ptr<float>() : nullptr, nullptr,
                              bs.empty() ? nullptr : btab.data_ptr(), otab.data_ptr(), work.data_ptr(), stream);
    release_plan(std::move(key), std::move(plan));
    check(rc, "svdq_merge");
    return outs;
}

// the mask tables both masked consumers need: combined mask bytes, their counts and one source start per work unit
struct MaskWalk {
    std::vector<at::Tensor> mb;
    at::Tensor mtab, ct, us, work;
};
static MaskWalk mask_walk(svdq_maskset *ms, const Plan &plan, at::TensorList masks, const std::vector<int64_t> &rows,
                          const c10::Device &dev, void *stream, const char *what) {
    MaskWalk m;
    const int64_t P = (int64_t)rows.size();
    TORCH_CHECK_VALUE((int64_t)masks.size() == P, what, ": one combined mask per parameter");
    for (int64_t p = 0; p < P; ++p) {
        TORCH_CHECK_VALUE(masks[p].device() == dev, what, ": all tensors must live on one device");
        TORCH_CHECK_VALUE(masks[p].numel() == rows[p], "Shape mismatch: tensor vs mask for parameter ", p);
        m.mb.push_back(mask_bytes(masks[p]));
    }
    m.work = bytes_on(dev, svdq_maskset_work_bytes(ms));
    m.ct = at::zeros({P}, at::TensorOptions().dtype(at::kLong).device(dev));
    m.mtab = table_of(m.mb, dev);
    m.us = at::empty({plan.sizes.n_units}, at::TensorOptions().dtype(at::kLong).device(dev));
    int rc = svdq_maskset_count_scan(ms, m.mtab.data_ptr(), m.ct.data_ptr<int64_t>(), nullptr, m.work.data_ptr(), stream);
    if (rc == SVDQ_OK)
        rc = svdq_maskset_unit_starts(ms, plan.h, m.mtab.data_ptr(), nullptr, m.ct.data_ptr<int64_t>(), m.work.data_ptr(),
                                      m.us.data_ptr<int64_t>(), stream);
    check(rc, what);
    return m;
}

// svdq_merge_masked on the buffers `compress_masked` / `compress_gather` returned for the same tensors, masks and
// settings: the merged delta of every parameter at FULL size (zeros where the mask is clear), + base when given
std::vector<at::Tensor> merge_masked(const at::Tensor &small, const at::Tensor &basis, const at::Tensor &mean,
                                     at::TensorList masks, int64_t n_tasks, double energy, int64_t max_rank, bool center,
                                     bool fp16, int64_t bits, int64_t stages, const at::Tensor &weights,
                                     at::TensorList base) {
    TORCH_CHECK(small.is_cuda() && basis.is_cuda(), "svdq operators take device tensors");
    const c10::Device dev = small.device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    const int64_t P = (int64_t)masks.size();
    TORCH_CHECK_VALUE(P >= 1, "Empty mask list");
    std::vector<int64_t> rows;
    for (int64_t p = 0; p < P; ++p) rows.push_back(masks[p].numel());
    TORCH_CHECK_VALUE(base.empty() || (int64_t)base.size() == P, "merge_masked: one base tensor per parameter, or none");
    at::Tensor w = weights.to(dev, at::kFloat).contiguous();
    TORCH_CHECK_VALUE((w.dim() == 1 || w.dim() == 2) && w.size(-1) == n_tasks,
                      "merge_masked: weights are [n_tasks] or [n_sets, n_tasks]");
    const int64_t n_sets = w.dim() == 2 ? w.size(0) : 1;
    svdq_maskset *ms = nullptr;
    check(svdq_maskset_create(&ms, (int32_t)P, rows.data()), "svdq_maskset_create");
    struct Guard {
        svdq_maskset *m;
        c10::Device d;
        ~Guard() {
            sync(d);
            svdq_maskset_destroy(m);
        }
    } ms_guard{ms, dev};
    PlanKey key{rows, n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan = acquire_plan(key, dev);
    check_artifacts("merge_masked", small, basis, mean, plan->sizes, center, dev);
    MaskWalk mw = mask_walk(ms, *plan, masks, rows, dev, stream, "merge_masked");
    std::vector<at::Tensor> outs, bs;
    for (int64_t p = 0; p < P; ++p) outs.push_back(floats_on(dev, rows[p]));
    for (size_t p = 0; p < base.size(); ++p) {
        TORCH_CHECK_VALUE(base[p].numel() == rows[p] && base[p].device() == dev, "merge_masked: base tensor ", p,
                          " does not match");
        bs.push_back(prep(base[p]));
    }
    at::Tensor otab = table_of(outs, dev), btab = bs.empty() ? at::Tensor() : table_of(bs, dev);
    at::Tensor share;
    if (n_sets > 1) share = at::full({n_sets}, 1.0 / (double)n_sets, at::TensorOptions().dtype(at::kFloat).device(dev));
    at::Tensor fill = at::ones({P}, at::TensorOptions().dtype(at::kInt).device(dev));      // no noise regions here
    at::Tensor work = bytes_on(dev, svdq_merge_work_bytes(plan->h, (int32_t)n_sets));
    const int rc = svdq_merge_masked(plan->h, mw.ct.data_ptr<int64_t>(), small.data_ptr(), basis.data_ptr(),
                                     (center && mean.numel() > 0) ? mean.data_ptr<float>() : nullptr, w.data_ptr<float>(),
                                     nullptr, (int32_t)n_sets, 0, n_sets > 1 ? share.data_ptr<float>() : nullptr, nullptr,
                                     mw.mtab.data_ptr(), mw.us.data_ptr<int64_t>(), fill.data_ptr<int32_t>(),
                                     bs.empty() ? nullptr : btab.data_ptr(), otab.data_ptr(), work.data_ptr(), stream);
    release_plan(std::move(key), std::move(plan));
    check(rc, "svdq_merge_masked");
    return outs;
}

// svdq_diagnostics / svdq_diagnostics_masked: [P, n_tasks, 6] float64 -- absolute_error, relative_error,
// max_absolute_error, mean_absolute_error, original_norm, reconstructed_norm of every (parameter, task) against the
// deltas the buffers were compressed from (masks empty: unmasked parameters)
at::Tensor diagnostics(at::TensorList deltas, at::TensorList masks, const at::Tensor &small, const at::Tensor &basis,
                       const at::Tensor &mean, int64_t n_tasks, double energy, int64_t max_rank, bool center, bool fp16,
                       int64_t bits, int64_t stages, bool add_mean) {
    std::vector<at::Tensor> vecs = prep_list(deltas, n_tasks, "diagnostics");
    const c10::Device dev = vecs[0].device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    std::vector<int64_t> rows = rows_of(vecs, n_tasks);
    const int64_t P = (int64_t)rows.size();
    TORCH_CHECK_VALUE(small.device() == dev && basis.device() == dev, "diagnostics: all tensors must live on one device");
    svdq_maskset *ms = nullptr;
    if (!masks.empty()) check(svdq_maskset_create(&ms, (int32_t)P, rows.data()), "svdq_maskset_create");
    struct Guard {
        svdq_maskset *m;
        c10::Device d;
        ~Guard() {
            if (m) {
                sync(d);
                svdq_maskset_destroy(m);
            }
        }
    } ms_guard{ms, dev};
    PlanKey key{rows, n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan = acquire_plan(key, dev);
    check_artifacts("diagnostics", small, basis, mean, plan->sizes, center, dev);
    at::Tensor table = table_of(vecs, dev);
    at::Tensor out = at::empty({P, n_tasks, 6}, at::TensorOptions().dtype(at::kDouble).device(dev));
    at::Tensor work = bytes_on(dev, svdq_diagnostics_work_bytes(plan->h));
    const float *mn = (center && mean.numel() > 0) ? mean.data_ptr<float>() : nullptr;
    int rc;
    if (ms) {
        MaskWalk mw = mask_walk(ms, *plan, masks, rows, dev, stream, "diagnostics");
        rc = svdq_diagnostics_masked(plan->h, table.data_ptr(), mw.mtab.data_ptr(), mw.us.data_ptr<int64_t>(),
                                     mw.ct.data_ptr<int64_t>(), small.data_ptr(), basis.data_ptr(), mn, add_mean ? 1 : 0,
                                     out.data_ptr<double>(), work.data_ptr(), stream);
        sync(dev);      // the mask tables die with this scope
    } else {
        svdq_small_layout L{};
        svdq_plan_small_layout(plan->h, &L);
        rc = svdq_diagnostics(plan->h, table.data_ptr(),
                              reinterpret_cast<const int64_t *>(small.data_ptr<uint8_t>() + L.rows_off), small.data_ptr(),
                              basis.data_ptr(), mn, add_mean ? 1 : 0, out.data_ptr<double>(), work.data_ptr(), stream);
        sync(dev);      // the pointer table dies with this scope
    }
    release_plan(std::move(key), std::move(plan));
    check(rc, "svdq_diagnostics");
    return out;
}

int64_t plan_cache_size() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return (int64_t)g_cache.size();
}

// ------------------------------------------------------------------------------------------ around the path
std::vector<at::Tensor> ingest(const at::Tensor &base, at::TensorList finetuned) {
    TORCH_CHECK_VALUE(!finetuned.empty(), "ingest: no fine-tuned tensors");
    const at::Tensor b = prep(base);
    const c10::Device dev = b.device();
    c10::DeviceGuard guard(dev);
    const int64_t n = b.numel(), N = (int64_t)finetuned.size();
    std::vector<at::Tensor> fts, out;
    for (const at::Tensor &f : finetuned) {
        TORCH_CHECK_VALUE(f.numel() == n, "ingest: fine-tuned tensor and base differ in size");
        fts.push_back(prep(f.to(dev)));
        out.push_back(floats_on(dev, n));
    }
    if (n == 0) {
        for (at::Tensor &o : out) o = o.view(base.sizes());
        return out;
    }
    svdq_config cfg{};
    cfg.energy_threshold = 0.9f;
    cfg.low_bits = 4;
    cfg.rtvq_stages = 2;
    cfg.fp16 = 1;
    auto plan = make_plan({n}, N, cfg, dev);
    at::Tensor tb = table_of({b}, dev), tf = table_of(fts, dev), td = table_of(out, dev);
    check(svdq_ingest(plan->h, tb.data_ptr(), tf.data_ptr(), td.data_ptr(), nullptr, stream_of(dev)), "svdq_ingest");
    sync(dev);                                         // the plan's device tables go away with it
    for (at::Tensor &o : out) o = o.view(base.sizes());
    return out;
}

at::Tensor task_gram(at::TensorList deltas, int64_t n_tasks) {
    std::vector<at::Tensor> vecs = prep_list(deltas, n_tasks, "task_gram");
    const c10::Device dev = vecs[0].device();
    c10::DeviceGuard guard(dev);
    svdq_config cfg{};
    cfg.energy_threshold = 0.9f;
    cfg.low_bits = 4;
    cfg.rtvq_stages = 2;
    cfg.fp16 = 1;
    auto plan = make_plan(rows_of(vecs, n_tasks), n_tasks, cfg, dev);
    at::Tensor table = table_of(vecs, dev);
    at::Tensor G = at::empty({n_tasks, n_tasks}, at::TensorOptions().dtype(at::kDouble).device(dev));
    check(svdq_task_gram(plan->h, table.data_ptr(), nullptr, plan->workspace.data_ptr(), G.data_ptr<double>(),
                         stream_of(dev)),
          "svdq_task_gram");
    sync(dev);
    return G;
}

// combine_masks + the flat[mask] index lists, batched over Q parameters (masks parameter-major, n_masks per parameter)
std::tuple<std::vector<at::Tensor>, std::vector<at::Tensor>, at::Tensor> mask_combine_indices(at::TensorList masks,
                                                                                              int64_t n_masks,
                                                                                              c10::string_view strategy) {
    TORCH_CHECK_VALUE(n_masks >= 1 && !masks.empty(), "Empty mask list");
    TORCH_CHECK_VALUE((int64_t)masks.size() % n_masks == 0, "mask_combine_indices: n_masks masks per parameter");
    int32_t code;
    if (strategy == "union") code = SVDQ_MASK_UNION;
    else if (strategy == "intersection") code = SVDQ_MASK_INTERSECTION;
    else if (strategy == "majority") code = SVDQ_MASK_MAJORITY;
    else TORCH_CHECK_VALUE(false, "Unknown mask strategy: ", std::string(strategy));
    const c10::Device dev = masks[0].device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    const int64_t Q = (int64_t)masks.size() / n_masks;
    std::vector<int64_t> numel(Q);
    std::vector<at::Tensor> flat, outs, idx;
    for (int64_t q = 0; q < Q; ++q) {
        numel[q] = masks[q * n_masks].numel();
        TORCH_CHECK_VALUE(numel[q] >= 1, "mask_combine_indices: empty mask");
        for (int64_t m = 0; m < n_masks; ++m) {
            const at::Tensor &t = masks[q * n_masks + m];
            TORCH_CHECK_VALUE(t.device() == dev, "mask_combine_indices: all masks must live on one device");
            TORCH_CHECK_VALUE(t.sizes() == masks[q * n_masks].sizes(), "Shape mismatch: mask ", t.sizes(), " vs mask ",
                              masks[q * n_masks].sizes());
            flat.push_back(mask_bytes(t));
        }
        outs.push_back(bytes_on(dev, numel[q]));
        idx.push_back(at::empty({numel[q]}, at::TensorOptions().dtype(at::kInt).device(dev)));
    }
    svdq_maskset *ms = nullptr;
    check(svdq_maskset_create(&ms, (int32_t)Q, numel.data()), "svdq_maskset_create");
    at::Tensor work = bytes_on(dev, svdq_maskset_work_bytes(ms));
    at::Tensor ct = at::zeros({Q}, at::TensorOptions().dtype(at::kLong).device(dev));
    at::Tensor mt = table_of(flat, dev), ot = table_of(outs, dev), it = table_of(idx, dev);
    const int rc = svdq_maskset_combine_indices(ms, mt.data_ptr(), (int32_t)n_masks, code, ot.data_ptr(), it.data_ptr(),
                                                nullptr, ct.data_ptr<int64_t>(), nullptr, work.data_ptr(), stream);
    sync(dev);                                         // the set's device tables go away with it
    svdq_maskset_destroy(ms);
    check(rc, "svdq_maskset_combine_indices");
    for (int64_t q = 0; q < Q; ++q) outs[q] = outs[q].view(at::kBool).view(masks[q * n_masks].sizes());
    return {outs, idx, ct};
}

struct BasisArgs {
    at::Tensor uh, ul, coef, mean;
    int64_t rows = 0, k = 0, nl = 0;
    bool fp16 = false;
    c10::Device dev{c10::kCUDA, 0};
};

BasisArgs basis_args(const at::Tensor &U_high, const at::Tensor &U_low, const at::Tensor &coef,
                     const c10::optional<at::Tensor> &mean, const char *what) {
    TORCH_CHECK(U_high.is_cuda() && U_low.is_cuda(), "svdq operators take device tensors");
    TORCH_CHECK_VALUE(U_high.dim() == 2 && U_low.dim() == 2 && U_high.size(0) == U_low.size(0), what,
                      ": U_high [D, k] and U_low [D, n_low]");
    BasisArgs a;
    a.dev = U_high.device();
    a.rows = U_high.size(0);
    a.k = U_high.size(1);
    a.nl = U_low.size(1);
    TORCH_CHECK_VALUE(a.k + a.nl <= 32, what, ": at most 32 basis columns");
    a.fp16 = (a.k ? U_high : U_low).scalar_type() == at::kHalf;
    const auto dt = a.fp16 ? at::kHalf : at::kFloat;
    a.uh = U_high.to(dt).contiguous();
    a.ul = U_low.to(a.dev, dt).contiguous();
    a.coef = coef.to(a.dev, at::kFloat).reshape({-1}).contiguous();
    TORCH_CHECK_VALUE(a.coef.numel() == a.k + a.nl, "Shape mismatch: ", a.coef.numel(), " coefficients for ", a.k + a.nl,
                      " basis columns");
    if (mean.has_value() && mean->defined()) {
        a.mean = prep(mean->to(a.dev));
        TORCH_CHECK_VALUE(a.mean.numel() == a.rows, what, ": mean has ", a.mean.numel(), " elements, the basis ", a.rows, " rows");
    }
    return a;
}

at::Tensor reconstruct(const at::Tensor &U_high, const at::Tensor &U_low, const at::Tensor &coef,
                       const c10::optional<at::Tensor> &mean, double scale) {
    BasisArgs a = basis_args(U_high, U_low, coef, mean, "reconstruct");
    c10::DeviceGuard guard(a.dev);
    at::Tensor out = floats_on(a.dev, a.rows);
    if (a.rows > 0)
        check(svdq_reconstruct(a.k ? a.uh.data_ptr() : nullptr, a.nl ? a.ul.data_ptr() : nullptr, a.fp16, a.rows,
                               (int32_t)a.k, (int32_t)a.nl, a.coef.data_ptr<float>(),
                               a.mean.defined() ? a.mean.data_ptr<float>() : nullptr, (float)scale, out.data_ptr<float>(),
                               stream_of(a.dev)),
              "svdq_reconstruct");
    return out;
}

at::Tensor recon_error(const at::Tensor &U_high, const at::Tensor &U_low, const at::Tensor &coef,
                       const c10::optional<at::Tensor> &mean, const at::Tensor &orig) {
    BasisArgs a = basis_args(U_high, U_low, coef, mean, "recon_error");
    c10::DeviceGuard guard(a.dev);
    at::Tensor x = prep(orig.to(a.dev));
    TORCH_CHECK_VALUE(x.numel() == a.rows && a.rows >= 1, "Shape mismatch: original has ", x.numel(), " elements, the basis ",
                      a.rows, " rows");
    at::Tensor out = at::empty({6}, at::TensorOptions().dtype(at::kDouble).device(a.dev));
    at::Tensor work = bytes_on(a.dev, svdq_recon_error_work_bytes(a.rows));
    check(svdq_recon_error(a.k ? a.uh.data_ptr() : nullptr, a.nl ? a.ul.data_ptr() : nullptr, a.fp16, a.rows, (int32_t)a.k,
                           (int32_t)a.nl, a.coef.data_ptr<float>(), a.mean.defined() ? a.mean.data_ptr<float>() : nullptr,
                           nullptr, x.data_ptr<float>(), out.data_ptr<double>(), work.data_ptr(), stream_of(a.dev)),
          "svdq_recon_error");
    return out;
}

}  // namespace

TORCH_LIBRARY(svdq, m) {
    m.def("rtvq_quantize(Tensor x, int bits, int stages) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("rtvq_dequantize(Tensor codes, Tensor scale, Tensor zero_point) -> Tensor");
    m.def("mask_combine(Tensor[] masks, str strategy) -> Tensor");
    m.def("mask_select(Tensor x, Tensor mask, bool invert) -> Tensor");
    m.def("compress(Tensor[] deltas, int n_tasks, float energy, int max_rank, bool center, bool fp16, int bits, "
          "int stages) -> (Tensor, Tensor, Tensor)");
    m.def("compress_masked(Tensor[] deltas, Tensor[] masks, int n_tasks, float energy, int max_rank, bool center, bool fp16, "
          "int bits, int stages) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("compress_gather(Tensor[] deltas, Tensor[] masks, int n_tasks, float energy, int max_rank, bool center, bool fp16, "
          "int bits, int stages) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("compress_from_base(Tensor[] finetuned, Tensor[] base, int n_tasks, float energy, int max_rank, bool center, "
          "bool fp16, int bits, int stages) -> (Tensor, Tensor, Tensor)");
    m.def("mask_combine_indices(Tensor[] masks, int n_masks, str strategy) -> (Tensor[], Tensor[], Tensor)");
    m.def("reconstruct(Tensor U_high, Tensor U_low, Tensor coef, Tensor? mean, float scale) -> Tensor");
    m.def("recon_error(Tensor U_high, Tensor U_low, Tensor coef, Tensor? mean, Tensor orig) -> Tensor");
    m.def("merge(Tensor small, Tensor basis, Tensor mean, int[] rows, int n_tasks, float energy, int max_rank, bool center, "
          "bool fp16, int bits, int stages, Tensor weights, Tensor[] base) -> Tensor[]");
    m.def("merge_masked(Tensor small, Tensor basis, Tensor mean, Tensor[] masks, int n_tasks, float energy, int max_rank, "
          "bool center, bool fp16, int bits, int stages, Tensor weights, Tensor[] base) -> Tensor[]");
    m.def("diagnostics(Tensor[] deltas, Tensor[] masks, Tensor small, Tensor basis, Tensor mean, int n_tasks, float energy, "
          "int max_rank, bool center, bool fp16, int bits, int stages, bool add_mean) -> Tensor");
    m.def("ingest(Tensor base, Tensor[] finetuned) -> Tensor[]");
    m.def("task_gram(Tensor[] deltas, int n_tasks) -> Tensor");
    m.def("plan_cache_size() -> int", plan_cache_size);
}

// "CUDA" is the HIP device key on ROCm builds of PyTorch; nothing is registered for CPU
TORCH_LIBRARY_IMPL(svdq, CUDA, m) {
    m.impl("rtvq_quantize", rtvq_quantize);
    m.impl("rtvq_dequantize", rtvq_dequantize);
    m.impl("mask_combine", mask_combine);
    m.impl("mask_select", mask_select);
    m.impl("compress", compress);
    m.impl("compress_masked", compress_masked);
    m.impl("compress_gather", compress_gather);
    m.impl("compress_from_base", compress_from_base);
    m.impl("mask_combine_indices", mask_combine_indices);
    m.impl("reconstruct", reconstruct);
    m.impl("recon_error", recon_error);
    m.impl("merge", merge);
    m.impl("merge_masked", merge_masked);
    m.impl("diagnostics", diagnostics);
    m.impl("ingest", ingest);
    m.impl("task_gram", task_gram);
}
