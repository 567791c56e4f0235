// HTF_POT_JIT: kernels generated at run time for a traced pair energy (hoomd_tf_amd/codegen.py -> csrc/jit_unit.hip ->
// `hipcc --genco`): the code object is loaded with the HIP module API and its twelve kernels -- the instantiations of the library's
// own row loops around the generated pair_eval body -- are launched with the arguments the built-in closed forms get.
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "box_math.h"
#include "htf_common.h"
#include "htf_internal.h"

namespace htf {

// libhiprtc, bound at run time like librccl in halo.hip: a box that never traces a model never loads it
struct Rtc {
    bool ok = false;
    decltype(&hiprtcCreateProgram) Create = nullptr;
    decltype(&hiprtcCompileProgram) Compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) LogSize = nullptr;
    decltype(&hiprtcGetProgramLog) Log = nullptr;
    decltype(&hiprtcGetCodeSize) CodeSize = nullptr;
    decltype(&hiprtcGetCode) Code = nullptr;
    decltype(&hiprtcDestroyProgram) Destroy = nullptr;
    decltype(&hiprtcGetErrorString) ErrorString = nullptr;
};

static Rtc &rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        if (const char *p = std::getenv("HTF_HIPRTC_LIB")) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.Create = (decltype(r.Create))dlsym(h, "hiprtcCreateProgram");
        r.Compile = (decltype(r.Compile))dlsym(h, "hiprtcCompileProgram");
        r.LogSize = (decltype(r.LogSize))dlsym(h, "hiprtcGetProgramLogSize");
        r.Log = (decltype(r.Log))dlsym(h, "hiprtcGetProgramLog");
        r.CodeSize = (decltype(r.CodeSize))dlsym(h, "hiprtcGetCodeSize");
        r.Code = (decltype(r.Code))dlsym(h, "hiprtcGetCode");
        r.Destroy = (decltype(r.Destroy))dlsym(h, "hiprtcDestroyProgram");
        r.ErrorString = (decltype(r.ErrorString))dlsym(h, "hiprtcGetErrorString");
        r.ok = r.Create && r.Compile && r.LogSize && r.Log && r.CodeSize && r.Code && r.Destroy;
    });
    return r;
}

struct JitKernels {
    hipModule_t mod = nullptr;
    hipFunction_t rows2[2][2] = {}; // [fp64 positions][tensor written]
    hipFunction_t row1v[2][2] = {};
    hipFunction_t tails4[2] = {};   // [tensor written]: fp32 positions, four rows per wave with merged tails
    hipFunction_t eval[2][2] = {};  // [fp64 tensor][virial]
    hipFunction_t train[2] = {};    // [fp64 tensor]: only in a unit compiled with weights (HTF_JIT_NPARAMS)
    hipFunction_t train_list[2] = {}; // [fp64 positions]: the same sweep from the index list
    int nparams = 0;
};

int jit_create(const void *image, size_t bytes, JitKernels **out) {
    HTF_REQUIRE(image && bytes > 0 && out, "HTF_POT_JIT: no code object (desc.jit_image)");
    JitKernels *k = new (std::nothrow) JitKernels();
    if (!k) {
        set_error("HTF_POT_JIT: out of host memory");
        return HTF_ERR_NOMEM;
    }
    hipError_t e = hipModuleLoadData(&k->mod, image);
    if (e != hipSuccess) {
        set_error("HTF_POT_JIT: hipModuleLoadData failed: %s (was the unit compiled for this GPU's architecture?)", hipGetErrorString(e));
        delete k;
        return HTF_ERR_DEVICE;
    }
    static const char *rows2[2][2] = {{"htf_jit_rows2_f32_nostore", "htf_jit_rows2_f32_store"}, {"htf_jit_rows2_f64_nostore", "htf_jit_rows2_f64_store"}};
    static const char *row1v[2][2] = {{"htf_jit_row1v_f32_nostore", "htf_jit_row1v_f32_store"}, {"htf_jit_row1v_f64_nostore", "htf_jit_row1v_f64_store"}};
    static const char *eval[2][2] = {{"htf_jit_eval_f32", "htf_jit_eval_f32_virial"}, {"htf_jit_eval_f64", "htf_jit_eval_f64_virial"}};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            if (e == hipSuccess) e = hipModuleGetFunction(&k->rows2[a][b], k->mod, rows2[a][b]);
            if (e == hipSuccess) e = hipModuleGetFunction(&k->row1v[a][b], k->mod, row1v[a][b]);
            if (e == hipSuccess) e = hipModuleGetFunction(&k->eval[a][b], k->mod, eval[a][b]);
        }
    if (e == hipSuccess) e = hipModuleGetFunction(&k->tails4[0], k->mod, "htf_jit_tails4_f32_nostore");
    if (e == hipSuccess) e = hipModuleGetFunction(&k->tails4[1], k->mod, "htf_jit_tails4_f32_store");
    if (e != hipSuccess) {
        set_error("HTF_POT_JIT: the code object lacks a kernel of csrc/jit_unit.hip: %s", hipGetErrorString(e));
        (void)hipModuleUnload(k->mod);
        delete k;
        return HTF_ERR_INVALID;
    }
    // optional: the training sweep and its parameter count (a unit traced with weights)
    hipDeviceptr_t np_ptr = nullptr;
    size_t np_bytes = 0;
    if (hipModuleGetGlobal(&np_ptr, &np_bytes, k->mod, "htf_jit_nparams") == hipSuccess && np_bytes == sizeof(int)) {
        if (hipMemcpy(&k->nparams, np_ptr, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) k->nparams = 0;
    } else {
        k->nparams = 0;
    }
    // (a unit with a row function reads weights without carrying a training sweep: such a model trains on the torch route)
    if (!(hipModuleGetFunction(&k->train[0], k->mod, "htf_jit_train_f32") == hipSuccess &&
          hipModuleGetFunction(&k->train[1], k->mod, "htf_jit_train_f64") == hipSuccess))
        k->train[0] = k->train[1] = nullptr;
    if (!(k->train[0] && hipModuleGetFunction(&k->train_list[0], k->mod, "htf_jit_train_list_f32") == hipSuccess &&
          hipModuleGetFunction(&k->train_list[1], k->mod, "htf_jit_train_list_f64") == hipSuccess))
        k->train_list[0] = k->train_list[1] = nullptr;
    (void)hipGetLastError();
    *out = k;
    return HTF_OK;
}

int jit_num_params(const JitKernels *k) { return k ? k->nparams : 0; }

void jit_destroy(JitKernels *k) {
    if (!k) return;
    if (k->mod) (void)hipModuleUnload(k->mod);
    delete k;
}

template <typename PT>
static int launch_fused_t(const PotParams &p, const void *pos, unsigned N, unsigned NN, unsigned offset, unsigned batch, const htf_box *hb,
                          const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax, void *force,
                          void *virial9, int out_f64, unsigned *check_count, float4 *positions_out, float4 *dest, unsigned *counts_io,
                          hipStream_t s) {
    constexpr int f64 = sizeof(PT) == 8 ? 1 : 0;
    BoxT<PT> b = make_boxt<PT>(hb);
    PT rc2 = (PT)rmax * (PT)rmax;
    PotParams pp = p;
    const int store = dest != nullptr ? 1 : 0;
    if (virial9 == nullptr) {
        void *args[] = {&pos, &N, &NN, &offset, &batch, &b, &n_neigh, &nlist, &head_list, &rc2, &force, &out_f64, &pp, &check_count,
                        &positions_out, &dest, &counts_io};
        static const char *tails_env = getenv("HTF_JIT_TAILS"); // "0": the two-row form at every size (A/B; forces then do not depend on how a step is cut)
        if (!f64 && batch >= 49152u && !(tails_env && tails_env[0] == '0')) {
            // launch_fused's rule for the closed forms BASELINE times: four rows per wave, their tails in one trip
            HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->tails4[store], ((batch + 3) / 4 + 3) / 4, 1, 1, 256, 1, 1, 0, s, args, nullptr));
            return HTF_OK;
        }
        const unsigned grid = ((batch + 1) / 2 + 3) / 4; // two rows per wave, four waves per workgroup (launch_fused's HTF_ROWS_LAUNCH)
        HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->rows2[f64][store], grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    } else {
        void *args[] = {&pos, &N, &NN, &offset, &batch, &b, &n_neigh, &nlist, &head_list, &rc2, &force, &virial9, &out_f64, &pp,
                        &check_count, &positions_out, &dest, &counts_io};
        HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->row1v[f64][store], (batch + 3) / 4, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    }
    return HTF_OK;
}

int jit_launch_fused(const PotParams &p, const void *pos, int pos_dtype, unsigned N, unsigned NN, unsigned offset, unsigned batch,
                     const htf_box *box, const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax,
                     void *force, void *virial9, int out_f64, unsigned *check_count, float4 *positions_out, float4 *dest,
                     unsigned *counts_io, hipStream_t s) {
    HTF_REQUIRE(p.jit, "HTF_POT_JIT: the potential has no kernels");
    if (pos_dtype == HTF_F32)
        return launch_fused_t<float>(p, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmax, force, virial9, out_f64, check_count,
                                     positions_out, dest, counts_io, s);
    return launch_fused_t<double>(p, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmax, force, virial9, out_f64, check_count,
                                  positions_out, dest, counts_io, s);
}

int jit_launch_eval(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force, void *virial9,
                    int out_f64, const unsigned *counts, hipStream_t s) {
    HTF_REQUIRE(p.jit, "HTF_POT_JIT: the potential has no kernels");
    PotParams pp = p;
    void *args[] = {&nlist, &B, &NN, &force, &virial9, &out_f64, &pp, &counts};
    const unsigned grid = (B + 15) / 16; // G = 16 lanes per row: 4 rows per wave, 4 waves per workgroup
    HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->eval[in_dtype == HTF_F64 ? 1 : 0][virial9 != nullptr ? 1 : 0], grid, 1, 1, 256, 1, 1, 0, s,
                                        args, nullptr));
    return HTF_OK;
}

int jit_launch_train(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN, const void *labels, int lab_f64,
                     void *pred, float *partials, unsigned grid, hipStream_t s) {
    HTF_REQUIRE(p.jit && p.jit->train[0] && p.jit->nparams > 0, "HTF_POT_JIT: this generated unit was compiled without weights: nothing to train");
    HTF_REQUIRE(p.theta != nullptr, "HTF_POT_JIT: a trainable traced energy needs its device parameter vector (desc.d_theta)");
    PotParams pp = p;
    void *args[] = {&nlist, &B, &NN, &labels, &lab_f64, &pred, &pp, &partials};
    HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->train[in_dtype == HTF_F64 ? 1 : 0], grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    return HTF_OK;
}

int jit_launch_train_list(const PotParams &p, const void *pos, int pos_dtype, unsigned B, unsigned NN, const htf_box *box,
                          const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax, const void *labels,
                          int lab_f64, void *pred, float *partials, unsigned grid, hipStream_t s) {
    HTF_REQUIRE(p.jit && p.jit->train_list[0] && p.jit->nparams > 0, "HTF_POT_JIT: this generated unit carries no list-form training sweep");
    HTF_REQUIRE(p.theta != nullptr, "HTF_POT_JIT: a trainable traced energy needs its device parameter vector (desc.d_theta)");
    PotParams pp = p;
    if (pos_dtype == HTF_F32) {
        BoxT<float> b = make_boxt<float>(box);
        float rc2 = (float)rmax * (float)rmax;
        void *args[] = {&pos, &B, &NN, &b, &n_neigh, &nlist, &head_list, &rc2, &labels, &lab_f64, &pred, &pp, &partials};
        HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->train_list[0], grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    } else {
        BoxT<double> b = make_boxt<double>(box);
        double rc2 = rmax * rmax;
        void *args[] = {&pos, &B, &NN, &b, &n_neigh, &nlist, &head_list, &rc2, &labels, &lab_f64, &pred, &pp, &partials};
        HTF_CHECK_HIP(hipModuleLaunchKernel(p.jit->train_list[1], grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    }
    return HTF_OK;
}

} // namespace htf

extern "C" int htf_jit_available(void) { return htf::rtc().ok ? 1 : 0; }

extern "C" int htf_jit_compile(const char *unit_source, const char *arch, int n_headers, const char *const *header_names,
                               const char *const *header_texts, int n_options, const char *const *options, void **image,
                               size_t *image_bytes, char *log, size_t log_bytes) {
    using namespace htf;
    HTF_REQUIRE(unit_source && arch && image && image_bytes && n_headers >= 0 && n_options >= 0 && (n_headers == 0 || (header_names && header_texts)) &&
                    (n_options == 0 || options),
                "htf_jit_compile: null pointer");
    if (log && log_bytes) log[0] = 0;
    *image = nullptr;
    *image_bytes = 0;
    Rtc &r = rtc();
    if (!r.ok) {
        set_error("htf_jit_compile: libhiprtc could not be loaded (HTF_HIPRTC_LIB names it explicitly)");
        return HTF_ERR_DEVICE;
    }
    hiprtcProgram prog = nullptr;
    hiprtcResult rc = r.Create(&prog, unit_source, "jit_unit.hip", n_headers, const_cast<const char **>(header_texts),
                               const_cast<const char **>(header_names));
    if (rc != HIPRTC_SUCCESS) {
        set_error("htf_jit_compile: hiprtcCreateProgram failed: %s", r.ErrorString ? r.ErrorString(rc) : "?");
        return HTF_ERR_DEVICE;
    }
    std::vector<const char *> opts;
    const std::string arch_opt = std::string("--offload-arch=") + arch;
    opts.push_back(arch_opt.c_str());
    for (int i = 0; i < n_options; ++i) opts.push_back(options[i]);
    rc = r.Compile(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    if (log && log_bytes > 1 && r.LogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        std::string text(ls, '\0');
        if (r.Log(prog, &text[0]) == HIPRTC_SUCCESS) {
            const size_t n = text.size() < log_bytes - 1 ? text.size() : log_bytes - 1;
            std::memcpy(log, text.data(), n);
            log[n] = 0;
        }
    }
    if (rc != HIPRTC_SUCCESS) {
        set_error("htf_jit_compile: hipRTC could not compile the generated unit: %s (see the log)", r.ErrorString ? r.ErrorString(rc) : "?");
        r.Destroy(&prog);
        return HTF_ERR_INVALID;
    }
    size_t cs = 0;
    void *buf = nullptr;
    if (r.CodeSize(prog, &cs) != HIPRTC_SUCCESS || cs == 0 || !(buf = std::malloc(cs)) || r.Code(prog, (char *)buf) != HIPRTC_SUCCESS) {
        std::free(buf);
        r.Destroy(&prog);
        set_error("htf_jit_compile: no code object came back");
        return HTF_ERR_DEVICE;
    }
    r.Destroy(&prog);
    *image = buf;
    *image_bytes = cs;
    return HTF_OK;
}

extern "C" void htf_jit_free(void *image) { std::free(image); }
