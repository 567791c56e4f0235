// HTF_POT_JIT: kernels generated at run time for a traced pair energy (hoomd_tf_amd/codegen.py -> csrc/jit_unit.hip ->
// `hipcc --genco`): the code object is loaded with the HIP module API and its twelve kernels -- the instantiations of the library's
// own row loops around the generated pair_eval body -- are launched with the arguments the built-in closed forms get.
#include "box_math.h"
#include "htf_common.h"
#include "htf_internal.h"

namespace htf {

struct JitKernels {
    hipModule_t mod = nullptr;
    hipFunction_t rows2[2][2] = {}; // [fp64 positions][tensor written]
    hipFunction_t row1v[2][2] = {};
    hipFunction_t eval[2][2] = {};  // [fp64 tensor][virial]
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
    if (e != hipSuccess) {
        set_error("HTF_POT_JIT: the code object lacks a kernel of csrc/jit_unit.hip: %s", hipGetErrorString(e));
        (void)hipModuleUnload(k->mod);
        delete k;
        return HTF_ERR_INVALID;
    }
    *out = k;
    return HTF_OK;
}

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

} // namespace htf
