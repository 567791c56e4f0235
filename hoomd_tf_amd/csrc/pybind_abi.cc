// pybind11 binding of the C ABI (include/htf_amd.h, include/htf_standin.h): BASELINE's "thin pybind11 C-ABI" for the Python
// host side.  One template turns every entry point into a Python callable -- pointers cross as integers (device pointers,
// addresses of ctypes structures), everything else as itself -- so the module carries no type the C header does not
// describe and needs no per-function glue: the list below is the header's symbol list (tests/test_abi.py checks it).
// hoomd_tf_amd/_lib.py uses it when it is built (HTF_BINDING=ctypes selects ctypes prototypes of the same library).
// The GIL is released around every call, as ctypes does: the few entry points that wait (check_nlist's read-back, the halo's
// end, destroy) do not stall other Python threads.
//   build: g++ -O2 -shared -fPIC pybind_abi.cc -I<repo>/include $(python -m pybind11 --includes) -L.. -lhtf_amd -o ../_htf_abi.so
#include <pybind11/pybind11.h>
#include <string>

#include <cstdint>
#include <type_traits>

#include "htf_amd.h"
#include "htf_standin.h"

namespace py = pybind11;

namespace {
// how a C parameter travels: any pointer as an address, scalars unchanged
template <class T> struct Wire { using type = T; static T from(T v) { return v; } };
template <class T> struct Wire<T *> {
    using type = std::uintptr_t;
    static T *from(std::uintptr_t v) { return reinterpret_cast<T *>(v); }
};

template <class R> struct Ret {
    template <class F> static R call(F &&f) { return f(); }
};
template <class T> struct Ret<T *> { // a returned pointer (htf_get_*_buffer) as an address; a C string as bytes
    template <class F> static py::object call(F &&f) {
        T *p = f();
        if constexpr (std::is_same<typename std::remove_cv<T>::type, char>::value)
            return p ? py::object(py::bytes(p)) : py::object(py::none());
        else
            return py::int_(reinterpret_cast<std::uintptr_t>(p));
    }
};
template <> struct Ret<void> {
    template <class F> static py::object call(F &&f) {
        f();
        return py::none();
    }
};

template <class R, class... A>
void bind(py::module &m, const char *name, R (*fn)(A...)) {
    m.def(name, [fn](typename Wire<A>::type... a) {
        return Ret<R>::call([&] {
            py::gil_scoped_release nogil; // (the C side never calls back into Python)
            return fn(Wire<A>::from(a)...);
        });
    });
}
} // namespace

#define HTF_ABI_FUNCTIONS(X) \
    X(htf_last_error) \
    X(htf_abi_version) \
    X(htf_device_count) \
    X(htf_potential_create) \
    X(htf_potential_destroy) \
    X(htf_build_pair_vectors) \
    X(htf_eval_forces) \
    X(htf_eval_forces_typed) \
    X(htf_jit_available) \
    X(htf_jit_compile) \
    X(htf_jit_free) \
    X(htf_fused_forces) \
    X(htf_build_eval_forces) \
    X(htf_eval_forces2) \
    X(htf_eval2_num_partials) \
    X(htf_build_eval_forces2) \
    X(htf_build_eval2_num_partials) \
    X(htf_reduce_partials) \
    X(htf_bias_combine) \
    X(htf_potential_num_params) \
    X(htf_train_scratch_floats) \
    X(htf_train_pair_grad) \
    X(htf_train_pair_grad_list) \
    X(htf_optimizer_step) \
    X(htf_optimizer_step_n) \
    X(htf_potential_refresh) \
    X(htf_add_virial) \
    X(htf_add_scalar4) \
    X(htf_copy_positions) \
    X(htf_energy_sum) \
    X(htf_copy3) \
    X(htf_positions_forces_radial) \
    X(htf_check_nlist) \
    X(htf_nlist_rinv) \
    X(htf_top_k) \
    X(htf_rdf_histogram) \
    X(htf_rdf_finalize) \
    X(htf_rbf_expansion) \
    X(htf_eds_update) \
    X(htf_wrap_vector) \
    X(htf_create) \
    X(htf_destroy) \
    X(htf_set_potential) \
    X(htf_resize) \
    X(htf_compute_forces) \
    X(htf_compute_forces_rows) \
    X(htf_get_nlist_buffer) \
    X(htf_reset_nlist_buffer) \
    X(htf_get_positions_buffer) \
    X(htf_get_virial_buffer) \
    X(htf_get_batch_capacity) \
    X(htf_halo_available) \
    X(htf_halo_unique_id) \
    X(htf_halo_create) \
    X(htf_halo_destroy) \
    X(htf_halo_exchange_begin) \
    X(htf_halo_exchange_end) \
    X(htf_halo_exchange_n) \
    X(htf_halo_allreduce_max_f32) \
    X(htf_halo_comm_info) \
    X(htf_profile_enable) \
    X(htf_profile_read) \
    X(htfs_nve_step) \
    X(htfs_max_displacement2) \
    X(htfs_check_displacement2) \
    X(htfs_build_nlist) \
    X(htfs_cell_sort) \
    X(htfs_gather4) \
    X(htfs_gather4_tagged) \
    X(htfs_gather4_tagged_live) \
    X(htfs_cell_index) \
    X(htfs_set_gate) \
    X(htfs_commit_rebuild) \
    X(htfs_rebuild_nlist) \
    X(htfs_rebuild_nlist_ghosts) \
    X(htfs_check_rebuild_nlist) \
    X(htfs_slab_classify) \
    X(htfs_key_sort16) \
    X(htfs_segment_copy) \
    X(htfs_brick_migrate_pack) \
    X(htfs_brick_migrate_merge) \
    X(htfs_brick_pack_halo) \
    X(htfs_brick_nve_halo) \
    X(htfs_brick_nve_halo_peer) \
    X(htfs_set_step_epilogue) \
    X(htfs_use_step_epilogue) \
    X(htfs_brick_row_slots) \
    X(htfs_shared_alloc) \
    X(htfs_shared_free) \
    X(htfs_ipc_export) \
    X(htfs_ipc_import) \
    X(htfs_ipc_close) \
    X(htfs_mailbox_push) \
    X(htfs_mailbox_pull) \
    X(htfs_mailbox_allreduce_max_f32) \
    X(htfs_brick_pack_halo_peer) \
    X(htfs_brick_unpack_halo)

PYBIND11_MODULE(_htf_abi, m) {
    m.doc() = "pybind11 binding of libhtf_amd.so's C ABI: pointers as integers";
    // a stale module (or library): this module's templates were instantiated from one header, the library it resolved at load
    // time may have been built from another
    if (htf_abi_version() != HTF_AMD_ABI_VERSION)
        throw pybind11::import_error("_htf_abi: built against ABI version " + std::to_string(HTF_AMD_ABI_VERSION) + ", libhtf_amd.so reports " +
                                     std::to_string(htf_abi_version()) + "; rebuild with `make -C hoomd_tf_amd/csrc pybind`");
    m.attr("abi_version") = HTF_AMD_ABI_VERSION;
#define X(fn) bind(m, #fn, &fn);
    HTF_ABI_FUNCTIONS(X)
#undef X
}
