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
    for (const at::Tensor &m : masks)
        TORCH_CHECK_VALUE(m.sizes() == masks[0].sizes(), "Shape mismatch: mask ", m.sizes(), " vs mask ", masks[0].sizes());
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

std::tuple<at::Tensor, at::Tensor, at::Tensor> compress(at::TensorList deltas, int64_t n_tasks, double energy,
                                                        int64_t max_rank, bool center, bool fp16, int64_t bits,
                                                        int64_t stages) {
    std::vector<at::Tensor> vecs = prep_list(deltas, n_tasks, "compress");
    const c10::Device dev = vecs[0].device();
    c10::DeviceGuard guard(dev);
    void *stream = stream_of(dev);
    PlanKey key{rows_of(vecs, n_tasks), n_tasks, max_rank, bits, stages, energy, center, fp16, (int)dev.index(), stream};

    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::unique_ptr<Plan> plan;
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
        if (it->first == key) {
            plan = std::move(it->second);
            g_cache.erase(it);
            break;
        }
    if (!plan) {
        svdq_config cfg{};
        cfg.energy_threshold = (float)energy;
        cfg.max_rank = max_rank > 0 ? (int32_t)max_rank : 0;
        cfg.center = center;
        cfg.fp16 = fp16;
        cfg.low_bits = (int32_t)bits;
        cfg.rtvq_stages = (int32_t)stages;
        plan = make_plan(key.rows, n_tasks, cfg, dev);
        while (g_cache.size() >= kPlanCacheMax) {
            sync(g_cache.front().second->dev);                        // its tables may still be in use
            g_cache.pop_front();
        }
    }
    // basis and mean in ONE allocation, the mean right behind the basis (pass-2 time depends on it: DESIGN.md section 5)
    const int64_t bb = plan->sizes.basis_bytes, gap = (bb + 255) / 256 * 256;
    const int64_t nm = center ? plan->sizes.mean_floats * 4 : 0;
    at::Tensor out = bytes_on(dev, gap + nm);
    at::Tensor basis = out.narrow(0, 0, bb);
    at::Tensor mean = center ? out.narrow(0, gap, nm).view(at::kFloat) : floats_on(dev, 0);
    at::Tensor small = at::zeros({plan->sizes.small_bytes}, at::TensorOptions().dtype(at::kByte).device(dev));
    at::Tensor table = table_of(vecs, dev);
    const int rc = svdq_compress(plan->h, table.data_ptr(), nullptr, plan->workspace.data_ptr(), small.data_ptr(),
                                 basis.data_ptr(), center ? mean.data_ptr<float>() : nullptr, stream);
    g_cache.emplace_back(std::move(key), std::move(plan));
    check(rc, "svdq_compress");
    // inputs and the table are only read by the kernels just enqueued; temporaries are released stream-ordered
    return {small, basis, mean};
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

}  // namespace

TORCH_LIBRARY(svdq, m) {
    m.def("rtvq_quantize(Tensor x, int bits, int stages) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("rtvq_dequantize(Tensor codes, Tensor scale, Tensor zero_point) -> Tensor");
    m.def("mask_combine(Tensor[] masks, str strategy) -> Tensor");
    m.def("mask_select(Tensor x, Tensor mask, bool invert) -> Tensor");
    m.def("compress(Tensor[] deltas, int n_tasks, float energy, int max_rank, bool center, bool fp16, int bits, "
          "int stages) -> (Tensor, Tensor, Tensor)");
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
    m.impl("ingest", ingest);
    m.impl("task_gram", task_gram);
}
