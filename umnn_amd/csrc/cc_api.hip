// C-ABI plumbing of libumnn_cc.so: argument validation, error reporting, introspection, and the
// host-side Clenshaw-Curtis tables (reference: models/UMNN/ParallelNeuralIntegral.py:14-34).
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "cc_host.h"

static thread_local char g_err[512] = "";
static thread_local const char* g_last_kernel = "";
static std::atomic<long long> g_launches{0};

int umnn_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int umnn_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

// last kernel per class of launch, process-wide (backward runs on autograd worker threads, so the thread-local
// g_last_kernel of the calling thread never sees it)
static std::atomic<const char*> g_last_of[3] = {{""}, {""}, {""}};
void umnn_note_launch(const char* kernel_name) {
    g_last_kernel = kernel_name;
    const int tag = !strncmp(kernel_name, "cc_bwd_dh", 9) || !strncmp(kernel_name, "cc_bwd_reduce", 13) ? UMNN_PROF_FINISH
                    : !strncmp(kernel_name, "cc_bwd", 6) ? UMNN_PROF_BACKWARD : UMNN_PROF_FORWARD;
    g_last_of[tag].store(kernel_name, std::memory_order_relaxed);
    g_launches.fetch_add(1, std::memory_order_relaxed);
}
// the conditioner's fused kernel keeps its own books: umnn_launch_count() counts QUADRATURE launches (tests assert
// "compute_ll = nb_flow launches", "invert = d launches per block" with it)
static std::atomic<long long> g_made_launches{0};
static std::atomic<const char*> g_last_made{""};
void umnn_note_made_launch(const char* kernel_name) {
    g_last_made.store(kernel_name, std::memory_order_relaxed);
    g_made_launches.fetch_add(1, std::memory_order_relaxed);
}
extern "C" long long umnn_made_launch_count(void) { return g_made_launches.load(std::memory_order_relaxed); }
extern "C" const char* umnn_last_made_kernel_name(void) { return g_last_made.load(std::memory_order_relaxed); }
extern "C" const char* umnn_last_kernel_name_of(int tag) {
    return tag >= 0 && tag <= 2 ? g_last_of[tag].load(std::memory_order_relaxed) : "";
}

int umnn_num_cus() {
    static std::atomic<int> cus[64];                       // per device ordinal; 0 = not queried yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (!v) {
        hipDeviceProp_t prop;
        v = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Flag words of the forward overflow protocol (cc_forward_bf16.hip): 256 zeroed 64-bit words per device, allocated on the first
// fp16-piece launch of that device (in relaxed capture mode, should that launch be recorded into a hipGraph); every launch takes
// the next generation number and the word generation % 256.
int umnn_ovf_slot(unsigned long long** flag, unsigned long long* gen) {
    static std::mutex mu;
    static unsigned long long* ring[64] = {nullptr};
    static std::atomic<unsigned long long> next{1};
    int dev = 0;
    if (int rc = umnn_check(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (dev < 0 || dev >= 64) return umnn_fail(UMNN_EINVAL, "device ordinal above 63");
    unsigned long long* r = __atomic_load_n(&ring[dev], __ATOMIC_ACQUIRE);
    if (!r) {
        std::lock_guard<std::mutex> lk(mu);
        r = ring[dev];
        if (!r) {
            hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
            (void)hipThreadExchangeStreamCaptureMode(&mode);
            hipError_t e = hipMalloc(&r, 256 * sizeof(unsigned long long));
            // (hipMemset of device memory is asynchronous to the host and runs on the legacy null stream, which non-blocking streams --
            // PyTorch's side streams -- are not ordered against: wait for it before the pointer is published)
            if (e == hipSuccess) e = hipMemset(r, 0, 256 * sizeof(unsigned long long));
            if (e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipThreadExchangeStreamCaptureMode(&mode);
            if (e != hipSuccess) return umnn_check(e, "overflow flag ring");
            __atomic_store_n(&ring[dev], r, __ATOMIC_RELEASE);
        }
    }
    const unsigned long long g = next.fetch_add(1, std::memory_order_relaxed);
    *flag = r + (g & 255);
    *gen = g;
    return 0;
}

int umnn_allow_lds(const void* fn, size_t bytes) {
    // raising the dynamic-LDS cap is per (device, function); cache what we already granted
    struct Key { int dev; const void* fn; bool operator==(const Key& o) const { return dev == o.dev && fn == o.fn; } };
    struct Hash { size_t operator()(const Key& k) const { return std::hash<const void*>()(k.fn) ^ ((size_t)k.dev * 0x9e3779b97f4a7c15ull); } };
    static std::mutex mu;
    static std::unordered_map<Key, size_t, Hash> granted;
    int dev = 0;
    if (int rc = umnn_check(hipGetDevice(&dev), "hipGetDevice")) return rc;
    std::lock_guard<std::mutex> lk(mu);
    auto it = granted.find(Key{dev, fn});
    if (it != granted.end() && it->second >= bytes) return 0;
    if (int rc = umnn_check(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"))
        return rc;
    granted[Key{dev, fn}] = bytes;
    return 0;
}

// ---- options: environment read once, atomics afterwards --------------------------------------------------------
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; }
static void load_env(UmnnOptions& o) {
    int fp = UMNN_PRECISION_F16X3, bp = UMNN_PRECISION_BF16X3;
    if (const char* ev = getenv("UMNN_FWD_PRECISION")) {
        if (!strcmp(ev, "fp32")) fp = UMNN_PRECISION_FP32;
        else if (!strcmp(ev, "bf16x6")) fp = UMNN_PRECISION_BF16X6;
        else if (!strcmp(ev, "bf16x3")) fp = UMNN_PRECISION_BF16X3;
    }
    if (const char* ev = getenv("UMNN_BWD_PRECISION")) bp = !strcmp(ev, "fp32") ? UMNN_PRECISION_FP32 : UMNN_PRECISION_BF16X3;
    o.fwd_precision = fp; o.bwd_precision = bp;
    int v = env_int("UMNN_FWD_P", -1); o.fwd_p = v == 1 || v == 2 ? v : -1;
    v = env_int("UMNN_FWD_NS", -1); o.fwd_ns = v == 1 || v == 2 || v == 4 ? v : -1;
    v = env_int("UMNN_FWD_TAIL", -1); o.fwd_tail = v < 0 ? -1 : (v != 0);
    // (every stored value must round-trip through umnn_set_option: out-of-range environment values fall back to the default)
    v = env_int("UMNN_FWD_PIPE", 1); o.fwd_pipe = v >= 0 && v <= 2 ? v : 1;   // 0 plain loop, 1 pipelined 16x16x32 loop, 2 the 32x32x16 formulation
    o.fwd_pad = env_int("UMNN_FWD_PAD", 1) != 0;
    v = env_int("UMNN_FWD_PAD_MIN", 1); o.fwd_pad_min = v >= 0 && v <= 127 ? v : 1;
    v = env_int("UMNN_BWD_NS", -1); o.bwd_ns = v >= 1 && v <= 32 ? v : -1;
    o.bwd_swp = env_int("UMNN_BWD_SWP", 1) != 0;
    o.bwd_ws = env_int("UMNN_BWD_WS", 1) != 0;
    v = env_int("UMNN_BWD_WS16", 1); o.bwd_ws16 = v >= 0 && v <= 2 ? v : 1;
    v = env_int("UMNN_FRONT_BWD2", 1); o.front_bwd2 = v >= 0 && v <= 2 ? v : 1;
}
UmnnOptions& umnn_options() {
    static UmnnOptions opts;
    static std::once_flag once;
    std::call_once(once, [] { load_env(opts); });
    return opts;
}
extern "C" int umnn_reload_env(void) { load_env(umnn_options()); return 0; }

static std::atomic<int>* option_slot(const char* name) {
    UmnnOptions& o = umnn_options();
    if (!name) return nullptr;
    if (!strcmp(name, "fwd_precision")) return &o.fwd_precision;
    if (!strcmp(name, "bwd_precision")) return &o.bwd_precision;
    if (!strcmp(name, "fwd_p")) return &o.fwd_p;
    if (!strcmp(name, "fwd_ns")) return &o.fwd_ns;
    if (!strcmp(name, "fwd_tail")) return &o.fwd_tail;
    if (!strcmp(name, "fwd_pipe")) return &o.fwd_pipe;
    if (!strcmp(name, "fwd_pad")) return &o.fwd_pad;
    if (!strcmp(name, "fwd_pad_min")) return &o.fwd_pad_min;
    if (!strcmp(name, "bwd_ns")) return &o.bwd_ns;
    if (!strcmp(name, "bwd_swp")) return &o.bwd_swp;
    if (!strcmp(name, "bwd_ws")) return &o.bwd_ws;
    if (!strcmp(name, "bwd_ws16")) return &o.bwd_ws16;
    if (!strcmp(name, "front_bwd2")) return &o.front_bwd2;
    return nullptr;
}
// the same per-option ranges load_env accepts (anything else would reach the launchers as "no such variant")
static bool option_value_ok(const char* name, int v) {
    if (!strcmp(name, "fwd_precision")) return v >= UMNN_PRECISION_FP32 && v <= UMNN_PRECISION_F16X3;
    if (!strcmp(name, "bwd_precision")) return v == UMNN_PRECISION_FP32 || v == UMNN_PRECISION_BF16X3;
    if (!strcmp(name, "fwd_p")) return v == -1 || v == 1 || v == 2;
    if (!strcmp(name, "fwd_ns")) return v == -1 || v == 1 || v == 2 || v == 4;
    if (!strcmp(name, "fwd_tail")) return v == -1 || v == 0 || v == 1;
    if (!strcmp(name, "fwd_pipe")) return v >= 0 && v <= 2;
    if (!strcmp(name, "fwd_pad")) return v == 0 || v == 1;
    if (!strcmp(name, "fwd_pad_min")) return v >= 0 && v <= 127;
    if (!strcmp(name, "bwd_ns")) return v == -1 || (v >= 1 && v <= 32);
    if (!strcmp(name, "bwd_swp")) return v == 0 || v == 1;
    if (!strcmp(name, "bwd_ws")) return v == 0 || v == 1;
    if (!strcmp(name, "bwd_ws16")) return v >= 0 && v <= 2;
    if (!strcmp(name, "front_bwd2")) return v >= 0 && v <= 2;
    return true;
}
extern "C" int umnn_set_option(const char* name, int value) {
    std::atomic<int>* s = option_slot(name);
    if (!s) return umnn_fail(UMNN_EINVAL, "umnn_set_option: unknown option name");
    if (!option_value_ok(name, value)) return umnn_fail(UMNN_EINVAL, "umnn_set_option: value out of range for this option");
    s->store(value, std::memory_order_relaxed);
    return 0;
}
extern "C" int umnn_get_option(const char* name, int* value) {
    std::atomic<int>* s = option_slot(name);
    if (!s || !value) return umnn_fail(UMNN_EINVAL, "umnn_get_option: unknown option name or null output");
    *value = s->load(std::memory_order_relaxed);
    return 0;
}
extern "C" int umnn_set_forward_precision(int mode) {
    if (mode < UMNN_PRECISION_FP32 || mode > UMNN_PRECISION_F16X3) return umnn_fail(UMNN_EINVAL, "unknown precision mode");
    umnn_options().fwd_precision = mode;
    return 0;
}
extern "C" int umnn_get_forward_precision(void) { return umnn_options().fwd_precision; }
extern "C" int umnn_set_backward_precision(int mode) {
    if (mode != UMNN_PRECISION_FP32 && mode != UMNN_PRECISION_BF16X3)
        return umnn_fail(UMNN_EINVAL, "backward precision must be UMNN_PRECISION_FP32 or UMNN_PRECISION_BF16X3");
    umnn_options().bwd_precision = mode;
    return 0;
}
extern "C" int umnn_get_backward_precision(void) { return umnn_options().bwd_precision; }

int umnn_check_io(const umnn_io* io) {
    if (!io) return 0;
    if ((io->x_dtype != UMNN_DTYPE_F32 && io->x_dtype != UMNN_DTYPE_BF16) || (io->h_dtype != UMNN_DTYPE_F32 && io->h_dtype != UMNN_DTYPE_BF16))
        return umnn_fail(UMNN_EINVAL, "umnn_io: dtype codes are UMNN_DTYPE_F32 or UMNN_DTYPE_BF16");
    return 0;
}

long long umnn_param_count(const umnn_mlp* net) {
    long long n = 0;
    for (int l = 0; l < net->n_linear; ++l) n += (long long)net->widths[l + 1] * net->widths[l] + net->widths[l + 1];
    return n;
}

int umnn_prepare_mlp(const umnn_mlp* net, int E, MlpDev* out, int* tmax, int* ksu) {
    if (!net) return umnn_fail(UMNN_EINVAL, "net is null");
    const int nl = net->n_linear;
    if (nl < 2 || nl > UMNN_MAX_LINEAR)
        return umnn_fail(UMNN_EUNSUPPORTED, "integrand MLP must have between 1 and UMNN_MAX_LINEAR-1 hidden layers");
    if (E < 0 || net->widths[0] != 1 + E)
        return umnn_fail(UMNN_EINVAL, "widths[0] must equal 1 + E (integration variable + embedding)");
    if (net->widths[nl] != 1) return umnn_fail(UMNN_EINVAL, "integrand output width must be 1");
    if (net->hidden_act != UMNN_ACT_LEAKY_RELU && net->hidden_act != UMNN_ACT_RELU)
        return umnn_fail(UMNN_EINVAL, "unknown hidden_act");
    if (net->out_act != UMNN_OUT_ELU_PLUS_ONE && net->out_act != UMNN_OUT_SIGMOID)
        return umnn_fail(UMNN_EINVAL, "unknown out_act");
    memset(out, 0, sizeof(*out));
    out->n_linear = nl;
    out->hidden_act = net->hidden_act;
    out->out_act = net->out_act;
    for (int l = 0; l <= nl; ++l) out->width[l] = net->widths[l];
    for (int l = 0; l < nl; ++l) {
        if (!net->W[l] || !net->b[l]) return umnn_fail(UMNN_EINVAL, "null weight or bias pointer");
        out->W[l] = net->W[l];
        out->b[l] = net->b[l];
    }
    const int L = nl - 1;
    int tm = 0, ks_common = -1;
    for (int l = 1; l <= L; ++l) {
        const int H = net->widths[l];
        if (H < 1 || H > UMNN_MAX_HIDDEN_WIDTH)
            return umnn_fail(UMNN_EUNSUPPORTED, "hidden width must be in [1, UMNN_MAX_HIDDEN_WIDTH]");
        out->t_out[l] = (H + 1 + 15) / 16;
        out->t_mfma[l] = out->t_out[l];
        out->ks_in[l] = (H + 1 + 3) / 4;
        if (out->t_out[l] > tm) tm = out->t_out[l];
        if (ks_common == -1) ks_common = out->ks_in[l];
        else if (ks_common != out->ks_in[l]) ks_common = 0;
    }
    int off = 0;
    for (int l = 1; l < L; ++l) {
        out->lds_off[l] = off;
        off += out->t_out[l + 1] * out->ks_in[l] * 64;
    }
    out->lds_off[L] = off;   // end of the images = start of kernel scratch
    // an "exact" variant additionally needs every layer to fill the same number of tiles
    for (int l = 1; l <= L; ++l)
        if (out->t_out[l] != tm) ks_common = 0;
    *tmax = tm;
    *ksu = ks_common > 0 ? ks_common : 0;
    return 0;
}

// ---- per-launch timing with hipEvents on the launch stream -------------------------------------
struct ProfRec { hipEvent_t a, b; double flops; int tag; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;
static thread_local hipEvent_t g_prof_open = nullptr;

void umnn_prof_begin(hipStream_t stream) {
    if (!g_prof_on) return;
    hipEvent_t a;
    if (hipEventCreate(&a) != hipSuccess) return;
    (void)hipEventRecord(a, stream);
    g_prof_open = a;
}

void umnn_prof_end(hipStream_t stream, double flops, int tag) {
    if (!g_prof_on || !g_prof_open) return;
    hipEvent_t b;
    if (hipEventCreate(&b) != hipSuccess) return;
    (void)hipEventRecord(b, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({g_prof_open, b, flops, tag});
    g_prof_open = nullptr;
}

extern "C" int umnn_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = on != 0;
    return 0;
}

static int profile_read(int tag, double* total_ms, long long* launches, double* total_flops) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0.0, fl = 0.0;
    long long cnt = 0;
    for (auto& r : g_prof) {
        if (tag >= 0 && r.tag != tag) continue;
        if (int rc = umnn_check(hipEventSynchronize(r.b), "hipEventSynchronize")) return rc;
        float t = 0.f;
        if (int rc = umnn_check(hipEventElapsedTime(&t, r.a, r.b), "hipEventElapsedTime")) return rc;
        ms += t;
        fl += r.flops;
        ++cnt;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    if (total_flops) *total_flops = fl;
    return 0;
}
extern "C" int umnn_profile_read(double* total_ms, long long* launches, double* total_flops) {
    return profile_read(-1, total_ms, launches, total_flops);
}
extern "C" int umnn_profile_read_tag(int tag, double* total_ms, long long* launches, double* total_flops) {
    if (tag < 0 || tag > 2) return umnn_fail(UMNN_EINVAL, "umnn_profile_read_tag: tag must be UMNN_PROF_FORWARD|BACKWARD|FINISH");
    return profile_read(tag, total_ms, launches, total_flops);
}

extern "C" const char* umnn_last_error(void) { return g_err; }
extern "C" int umnn_version(void) { return 100; }
extern "C" long long umnn_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
extern "C" const char* umnn_last_kernel_name(void) { return g_last_kernel; }

extern "C" double umnn_cc_forward_flops_per_integral(const umnn_mlp* net, int nb_steps) {
    if (!net || net->n_linear < 2) return 0.0;
    const int L = net->n_linear - 1;
    double node = net->widths[1] + net->widths[L];
    for (int l = 1; l < L; ++l) node += (double)net->widths[l] * net->widths[l + 1];
    const double once = (double)(net->widths[0] - 1) * net->widths[1];
    return 2.0 * ((nb_steps + 1) * node + once);
}

extern "C" int umnn_cc_tables_host(int nb_steps, float* w_host, float* s_host) {
    if (nb_steps < 1 || !w_host || !s_host) return umnn_fail(UMNN_EINVAL, "tables: nb_steps >= 1, non-null outputs");
    const int n = nb_steps;
    std::vector<double> Wj(n + 1);
    for (int j = 0; j <= n; ++j) Wj[j] = (j % 2) ? 0.0 : 2.0 / (1.0 - (double)j * j);
    Wj[0] = 1.0;
    for (int k = 0; k <= n; ++k) {
        // w_k = sum_j lam[j][k] * W_j with lam[j][k] = cos(jk pi/n) * 2/n, column 0 := .5*2/n, column n halved
        double acc = 0.0;
        for (int j = 0; j <= n; ++j) {
            double lam = std::cos((double)j * k * M_PI / n);
            if (k == 0) lam = .5;
            else if (k == n) lam = .5 * lam;
            acc += lam * 2 / n * Wj[j];
        }
        w_host[k] = (float)acc;
        s_host[k] = (float)std::cos((double)k * M_PI / n);
    }
    return 0;
}
