// HOOMD-blue 2.x side of the drop-in: the ForceCompute that takes the place of
// TensorflowCompute<M> / TensorflowComputeGPU (htf/TensorflowCompute.h:75-298, .cc:29-614) and forwards
// the per-step work to libhtf_amd.so (include/htf_amd.h).  Same constructor arguments, same exported
// methods (htf/TensorflowCompute.cc:422-486), same half-step hook (TensorflowCompute.h:53-71).
//
// What is different by design:
//  * get*Buffer() return raw DEVICE POINTERS (int64) of the context-owned side buffers, not CommStruct
//    addresses: there is no TF op to hand a descriptor to; shapes are [getBatchCapacity(), NN, 4] etc.
//  * a declarative model is lowered ONCE on the Python side to an htf_potential (hoomd_tf_amd.ops.Potential,
//    created through the same libhtf_amd.so) and installed with setPotential(handle); from then on
//    computeForces() is one C call per batch and Python is not in the step loop.  Without a potential the
//    compute falls back to the reference's protocol: build the pair vectors, call py_self._finish_update(batch).
//
// This file compiles against HOOMD-blue 2.x headers; in this repository (no HOOMD in the image) it is compiled
// against integration/hoomd_stub/ by tests/test_shim_compiles.py -- a syntax / type / signature check, not a run.
#pragma once
#include <hoomd/ForceCompute.h>
#include <hoomd/HalfStepHook.h>
#include <hoomd/md/NeighborList.h>
#include <pybind11/pybind11.h>

#include <stdexcept>

#include "htf_amd.h"

namespace hoomd_tf_amd {

//! TensorflowCompute.h:44-48
enum class FORCE_MODE { tf2hoomd, hoomd2tf };

//! TensorflowCompute.h:53-71: lets the integrator call the compute at the half step (hoomd2tf mode)
template <class T>
class HalfStepHookWrapper : public HalfStepHook {
public:
    T &m_f;
    explicit HalfStepHookWrapper(T &f) : m_f(f) {}
    void update(unsigned int timestep) override { m_f.computeForces(timestep); }
    void setSystemDefinition(std::shared_ptr<SystemDefinition>) override {}
};

class TensorflowComputeAMD : public ForceCompute {
public:
    //! TensorflowCompute.h:82-89, pybind .cc:431-438
    TensorflowComputeAMD(pybind11::object &py_self, std::shared_ptr<SystemDefinition> sysdef,
                         std::shared_ptr<NeighborList> nlist, Scalar r_cut, unsigned int nneighs, FORCE_MODE force_mode,
                         unsigned int period, unsigned int batch_size);
    TensorflowComputeAMD() = delete;
    virtual ~TensorflowComputeAMD();

    //! "tensorflow" log quantity, .cc:376-395
    std::vector<std::string> getProvidedLogQuantities() override { return {m_log_name}; }
    Scalar getLogValue(const std::string &quantity, unsigned int timestep) override;

    //! .cc:398-407 -- device pointers of the side buffers (see the header comment)
    int64_t getForcesBuffer() const;
    int64_t getPositionsBuffer() const { return reinterpret_cast<int64_t>(htf_get_positions_buffer(m_ctx)); }
    int64_t getBoxBuffer() const { return reinterpret_cast<int64_t>(&m_box); } // host: 3 x 3 doubles lo / hi / tilt
    int64_t getVirialBuffer() const { return reinterpret_cast<int64_t>(htf_get_virial_buffer(m_ctx)); }
    int64_t getNlistBuffer() const { return reinterpret_cast<int64_t>(htf_get_nlist_buffer(m_ctx)); }
    unsigned int getBatchCapacity() const { return htf_get_batch_capacity(m_ctx); }

    bool isDoublePrecision() const { return sizeof(Scalar) == 8; } // TensorflowCompute.h:117-124

    //! .cc:409-413
    void setMappedNlist(bool mn, unsigned int cg_typeid_start) {
        m_b_mapped_nlist = mn;
        m_cg_typeid_start = cg_typeid_start;
    }

    //! host copies of the side buffers, .cc:409-420 (tests and notebooks read these)
    std::vector<Scalar4> getForcesArray() const;
    std::vector<Scalar4> getNlistArray() const;     // fp32 on the device, widened to Scalar here
    std::vector<Scalar4> getPositionsArray() const; // type un-stuffed, as the model sees it
    std::vector<Scalar3> getBoxArray() const;
    std::vector<Scalar> getVirialArray() const;

    void computeForces(unsigned int timestep) override; // public: the half-step hook calls it (.h:144)

    unsigned int getVirialPitch() const { return m_virial.getPitch(); }
    std::shared_ptr<HalfStepHook> getHook() { return hook; }
    void addReferenceForce(std::shared_ptr<ForceCompute> force) { m_ref_forces.push_back(force); }

    // ---- the lowered model (no counterpart upstream: there the model is a TF graph behind _finish_update)
    //! install the potential `hoomd_tf_amd.ops.Potential(...).handle` (borrowed; Python keeps it alive)
    void setPotential(int64_t potential_handle, bool virial, bool check_nlist, int fused);
    //! hoomd2tf training: device parameter vector (Keras get_weights() order), optimizer rule and state
    void setTraining(int64_t d_theta, unsigned int n_params, int64_t d_opt_state, int opt_kind, float lr, float beta1,
                     float beta2, float epsilon, unsigned int nonneg_mask, float l1_reg0);

    pybind11::object m_py_self; //!< tensorflowcompute.py object: _start_update / _finish_update callbacks
    std::shared_ptr<HalfStepHookWrapper<TensorflowComputeAMD>> hook;

protected:
    virtual void reallocate();       //!< MaxParticleNumberChange, .cc:88-121
    void sumReferenceForces();       //!< .cc:250-269 -> labels
    void trainOnBatch(unsigned int offset, unsigned int n);
    void updateBox();                //!< .cc:271-282
    void fillArrays(htf_hoomd_arrays &a, const ArrayHandle<Scalar4> &pos, const ArrayHandle<unsigned int> &n_neigh,
                    const ArrayHandle<unsigned int> &nl, const ArrayHandle<unsigned int> &head,
                    const ArrayHandle<Scalar4> &force, const ArrayHandle<Scalar> &virial) const;
    void check(int rc) const;
    void recreateContext();

    std::shared_ptr<NeighborList> m_nlist;
    Scalar m_r_cut;
    unsigned int m_nneighs;
    FORCE_MODE m_force_mode;
    unsigned int m_period;
    unsigned int m_batch_size;
    bool m_b_mapped_nlist = false;
    unsigned int m_cg_typeid_start = 0;
    std::string m_log_name = "tensorflow"; // .cc:62
    std::vector<std::shared_ptr<ForceCompute>> m_ref_forces;

    htf_ctx *m_ctx = nullptr;
    const htf_potential *m_pot = nullptr;
    htf_config m_cfg{};
    double m_box[9] = {0};
    // training state (device pointers owned by the Python side)
    float *m_theta = nullptr, *m_opt_state = nullptr;
    unsigned int m_n_params = 0;
    htf_optimizer_desc m_opt{};
    void *m_labels = nullptr;   // Scalar4[maxN], device (owned)
    float *m_accum = nullptr;   // 1 + P floats, device (owned)
    float *m_scratch = nullptr; // htf_train_scratch_floats, device (owned)
    size_t m_scratch_floats = 0;
};

void export_TensorflowComputeAMD(pybind11::module &m);

} // namespace hoomd_tf_amd
