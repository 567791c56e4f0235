// See TensorflowComputeAMD.h.  Line references are into hoomd-tf v2.4.0 (htf/TensorflowCompute.cc).
#include "TensorflowComputeAMD.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>

namespace hoomd_tf_amd {

namespace {
constexpr int kScalar = sizeof(Scalar) == 8 ? HTF_F64 : HTF_F32;
// HOOMD-blue 2.x enqueues its kernels on the default stream; so does the plugin, and stream order is all
// the ordering the two need (the reference additionally issues a device-wide synchronize per batch, :208-211)
const htf_stream kHoomdStream = nullptr;

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("htf_amd shim: ") + what + ": " + hipGetErrorString(e));
}
} // namespace

TensorflowComputeAMD::TensorflowComputeAMD(pybind11::object &py_self, std::shared_ptr<SystemDefinition> sysdef,
                                           std::shared_ptr<NeighborList> nlist, Scalar r_cut, unsigned int nneighs,
                                           FORCE_MODE force_mode, unsigned int period, unsigned int batch_size)
    : ForceCompute(sysdef), m_py_self(py_self),
      hook(std::make_shared<HalfStepHookWrapper<TensorflowComputeAMD>>(*this)), m_nlist(nlist), m_r_cut(r_cut),
      m_nneighs(nneighs), m_force_mode(force_mode), m_period(period), m_batch_size(batch_size) {
    m_exec_conf->msg->notice(2) << "Starting TensorflowComputeAMD" << std::endl;
    if (htf_abi_version() != HTF_AMD_ABI_VERSION) // this plugin was compiled against another include/htf_amd.h than the library
        throw std::runtime_error("TensorflowComputeAMD: compiled against htf_amd ABI version " + std::to_string(HTF_AMD_ABI_VERSION) +
                                 ", libhtf_amd.so reports " + std::to_string(htf_abi_version()));
    m_cfg.r_cut = r_cut;
    m_cfg.nneighs = nneighs;
    m_cfg.force_mode = force_mode == FORCE_MODE::tf2hoomd ? HTF_TF2HOOMD : HTF_HOOMD2TF;
    m_cfg.period = period;
    m_cfg.batch_size = batch_size;
    m_cfg.scalar_dtype = kScalar; // isDoublePrecision()
    m_cfg.check_nlist = 0;
    m_cfg.virial = 0;
    m_cfg.max_n = m_pdata->getMaxN();
    m_cfg.fused = 2; // one kernel writes the pair-vector tensor and evaluates it (closed-form potentials)
    check(htf_create(&m_cfg, &m_ctx));
    // (the reference also raises pdata_flag::pressure_tensor here, :63-70; HOOMD asks computes for the
    //  virial through the same flag, nothing for the plugin to do beyond filling m_virial)
    if (m_nneighs > 0 && m_nlist->getStorageMode() == NeighborList::half) { // :73-84
        m_nlist->setStorageMode(NeighborList::full);
        m_exec_conf->msg->notice(8) << "Swapping to full neighbor list" << std::endl;
    }
    m_pdata->getMaxParticleNumberChangeSignal().connect<TensorflowComputeAMD, &TensorflowComputeAMD::reallocate>(this);
}

TensorflowComputeAMD::~TensorflowComputeAMD() {
    m_pdata->getMaxParticleNumberChangeSignal().disconnect<TensorflowComputeAMD, &TensorflowComputeAMD::reallocate>(this);
    htf_destroy(m_ctx);
    if (m_labels) (void)hipFree(m_labels);
    if (m_accum) (void)hipFree(m_accum);
    if (m_scratch) (void)hipFree(m_scratch);
}

void TensorflowComputeAMD::check(int rc) const {
    if (rc == HTF_OK) return;
    m_exec_conf->msg->error() << "htf_amd: " << htf_last_error() << std::endl;
    // the reference surfaces these as tf.errors.InvalidArgumentError through pybind (simmodel.py:195,223-224);
    // pybind11 turns std::invalid_argument into ValueError and std::runtime_error into RuntimeError
    if (rc == HTF_ERR_INVALID) throw std::invalid_argument(htf_last_error());
    throw std::runtime_error(htf_last_error());
}

void TensorflowComputeAMD::recreateContext() {
    htf_destroy(m_ctx);
    m_ctx = nullptr;
    m_cfg.max_n = m_pdata->getMaxN();
    check(htf_create(&m_cfg, &m_ctx));
    if (m_pot) check(htf_set_potential(m_ctx, m_pot));
}

void TensorflowComputeAMD::reallocate() { // :91-121
    check(htf_resize(m_ctx, m_pdata->getMaxN()));
    if (m_labels) {
        (void)hipFree(m_labels);
        m_labels = nullptr; // re-made at the next training step
    }
}

void TensorflowComputeAMD::setPotential(int64_t handle, bool virial, bool check_nlist, int fused) {
    m_pot = reinterpret_cast<const htf_potential *>(handle);
    if (bool(m_cfg.virial) != virial || bool(m_cfg.check_nlist) != check_nlist || m_cfg.fused != fused) {
        m_cfg.virial = virial; // SimModel(virial=..., check_nlist=...) simmodel.py:15
        m_cfg.check_nlist = check_nlist;
        m_cfg.fused = fused;
        recreateContext();
    } else {
        check(htf_set_potential(m_ctx, m_pot));
    }
}

void TensorflowComputeAMD::setTraining(int64_t d_theta, unsigned int n_params, int64_t d_opt_state, int opt_kind, float lr,
                                       float beta1, float beta2, float epsilon, unsigned int nonneg_mask, float l1_reg0) {
    m_theta = reinterpret_cast<float *>(d_theta);
    m_opt_state = reinterpret_cast<float *>(d_opt_state);
    m_n_params = n_params;
    m_opt = htf_optimizer_desc{};
    m_opt.kind = opt_kind;
    m_opt.lr = lr;
    m_opt.beta1 = beta1;
    m_opt.beta2 = beta2;
    m_opt.epsilon = epsilon;
    m_opt.nonneg_mask = nonneg_mask;
    m_opt.l1_reg[0] = l1_reg0;
    if (m_accum) (void)hipFree(m_accum);
    hip_check(hipMalloc((void **)&m_accum, (1 + (size_t)n_params) * sizeof(float)), "hipMalloc(accum)");
}

void TensorflowComputeAMD::updateBox() { // :271-282
    const BoxDim &box = m_pdata->getBox();
    const Scalar3 lo = box.getLo(), hi = box.getHi();
    const double b[9] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, box.getTiltFactorXY(), box.getTiltFactorXZ(), box.getTiltFactorYZ()};
    std::copy(b, b + 9, m_box);
}

void TensorflowComputeAMD::fillArrays(htf_hoomd_arrays &a, const ArrayHandle<Scalar4> &pos,
                                      const ArrayHandle<unsigned int> &n_neigh, const ArrayHandle<unsigned int> &nl,
                                      const ArrayHandle<unsigned int> &head, const ArrayHandle<Scalar4> &force,
                                      const ArrayHandle<Scalar> &virial) const {
    a = htf_hoomd_arrays{};
    a.pos = pos.data;
    a.N = m_pdata->getN();
    a.n_ghost = m_pdata->getNGhosts();
    a.n_neigh = n_neigh.data;
    a.nlist = nl.data;
    a.head_list = head.data;
    const BoxDim &box = m_pdata->getBox();
    const uchar3 per = box.getPeriodic();
    for (int d = 0; d < 3; ++d) {
        a.box.lo[d] = m_box[d];
        a.box.hi[d] = m_box[3 + d];
        a.box.tilt[d] = m_box[6 + d];
    }
    a.box.periodic[0] = per.x;
    a.box.periodic[1] = per.y;
    a.box.periodic[2] = per.z;
    a.force = force.data;
    a.virial = virial.data;
    a.virial_pitch = m_virial.getPitch();
}

void TensorflowComputeAMD::sumReferenceForces() { // :250-269, on the device (htf_gpu_add_scalar4, .cu:11-39)
    const unsigned int N = m_pdata->getN();
    hip_check(hipMemsetAsync(m_labels, 0, (size_t)N * sizeof(Scalar4), (hipStream_t)kHoomdStream), "hipMemsetAsync(labels)");
    for (auto const &f : m_ref_forces) {
        ArrayHandle<Scalar4> src(f->getForceArray(), access_location::device, access_mode::read);
        check(htf_add_scalar4(m_labels, src.data, kScalar, N, kHoomdStream));
    }
}

void TensorflowComputeAMD::trainOnBatch(unsigned int offset, unsigned int n) {
    // model.train_on_batch(x = this batch's inputs, y = labels[offset : offset + n]) (tensorflowcompute.py:366-370)
    // for a lowered trainable potential: prediction, MSE over the [n, 4] columns and d loss / d theta in ONE
    // sweep, the optimizer rule on the device, operand images rebuilt from theta.  No Python, no host copy.
    if (!m_pot || !m_theta) throw std::runtime_error("hoomd2tf: no trainable potential installed (setPotential / setTraining)");
    const size_t need = htf_train_scratch_floats(m_pot, n, m_nneighs);
    if (need > m_scratch_floats) {
        if (m_scratch) (void)hipFree(m_scratch);
        hip_check(hipMalloc((void **)&m_scratch, need * sizeof(float)), "hipMalloc(scratch)");
        m_scratch_floats = need;
    }
    const char *labels = static_cast<const char *>(m_labels) + (size_t)offset * sizeof(Scalar4);
    check(htf_train_pair_grad(m_pot, htf_get_nlist_buffer(m_ctx), HTF_F32, n, m_nneighs, labels, kScalar, nullptr, m_accum,
                              m_scratch, kHoomdStream));
    // (under MPI: one ncclAllReduce / MPI_Allreduce of m_accum[0 .. 1 + P) and of n here keeps the ranks' weights equal)
    const float scale = 1.0f / (4.0f * (float)n);
    if (m_n_params <= 8)
        check(htf_optimizer_step(m_theta, m_n_params, m_accum, scale, m_opt_state, &m_opt, kHoomdStream));
    else
        check(htf_optimizer_step_n(m_theta, m_n_params, m_accum, scale, m_opt_state, &m_opt, kHoomdStream));
    check(htf_potential_refresh(const_cast<htf_potential *>(m_pot), kHoomdStream));
}

void TensorflowComputeAMD::computeForces(unsigned int timestep) { // :129-216
    if (timestep % m_period != 0) return;
    if (m_batch_size == 0 && m_b_mapped_nlist) m_py_self.attr("_start_update")(); // startUpdate, :228-241
    if (m_prof) m_prof->push("TensorflowCompute");
    if (m_nneighs > 0) {
        if (m_nlist->getStorageMode() == NeighborList::half) { // :156-160
            m_exec_conf->msg->error() << "Must have full neighbor list" << std::endl;
            throw std::runtime_error("neighbor list wrong type");
        }
        m_nlist->compute(timestep); // :162-163
    }
    updateBox();
    const unsigned int N = m_pdata->getN();
    const bool training = m_force_mode == FORCE_MODE::hoomd2tf;
    if (training) { // labels, once per step for all batches (:177-187)
        if (!m_labels) hip_check(hipMalloc(&m_labels, (size_t)std::max(1u, m_pdata->getMaxN()) * sizeof(Scalar4)), "hipMalloc(labels)");
        if (m_ref_forces.empty()) {
            ArrayHandle<Scalar4> net(m_pdata->getNetForce(), access_location::device, access_mode::read);
            hip_check(hipMemcpyAsync(m_labels, net.data, (size_t)N * sizeof(Scalar4), hipMemcpyDeviceToDevice,
                                     (hipStream_t)kHoomdStream), "hipMemcpyAsync(labels)");
        } else {
            sumReferenceForces();
        }
    }
    {
        ArrayHandle<Scalar4> pos(m_pdata->getPositions(), access_location::device, access_mode::read);
        ArrayHandle<unsigned int> n_neigh(m_nlist->getNNeighArray(), access_location::device, access_mode::read);
        ArrayHandle<unsigned int> nl(m_nlist->getNListArray(), access_location::device, access_mode::read);
        ArrayHandle<unsigned int> head(m_nlist->getHeadList(), access_location::device, access_mode::read);
        ArrayHandle<Scalar4> force(m_force, access_location::device, training ? access_mode::read : access_mode::overwrite);
        ArrayHandle<Scalar> virial(m_virial, access_location::device, access_mode::readwrite);
        htf_hoomd_arrays a;
        fillArrays(a, pos, n_neigh, nl, head, force, virial);
        if (m_pot && !training) {
            // the whole step -- batch loop, pair vectors, evaluation into m_force, virial fold-in -- is this call
            check(htf_compute_forces(m_ctx, timestep, &a, kHoomdStream));
        } else {
            // no lowered potential (a generic model: Python evaluates it on the zero-copy side buffers), or a
            // training step: batch by batch, as the reference drives _finish_update (:143-206)
            const unsigned int bs = m_batch_size == 0 ? N : m_batch_size;
            for (unsigned int i = 0; bs > 0 && i < N / bs + 1; ++i) {
                const unsigned int offset = i * bs;
                if (offset >= N) break;
                const unsigned int n = std::min(N - offset, bs);
                check(htf_compute_forces_rows(m_ctx, timestep, &a, offset, n, kHoomdStream)); // stages nlist + positions
                if (training && m_theta)
                    trainOnBatch(offset, n);
                else
                    m_py_self.attr("_finish_update")(i); // finishUpdate, :218-226
            }
        }
    }
    if (m_prof) m_prof->pop();
}

Scalar TensorflowComputeAMD::getLogValue(const std::string &quantity, unsigned int timestep) { // :376-395
    if (quantity == m_log_name) {
        compute(timestep);
        return calcEnergySum();
    }
    m_exec_conf->msg->error() << "tensorflow:" << quantity << " is not a valid log quantity" << std::endl;
    throw std::runtime_error("Error getting log value");
}

int64_t TensorflowComputeAMD::getForcesBuffer() const {
    ArrayHandle<Scalar4> force(m_force, access_location::device, access_mode::read);
    return reinterpret_cast<int64_t>(force.data); // the plugin writes HOOMD's m_force in place (:107)
}

namespace {
template <class T>
std::vector<T> from_device(const void *d, size_t n) {
    std::vector<T> out(n);
    if (n && d) hip_check(hipMemcpy(out.data(), d, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    return out;
}
} // namespace

std::vector<Scalar4> TensorflowComputeAMD::getForcesArray() const {
    ArrayHandle<Scalar4> force(m_force, access_location::device, access_mode::read);
    return from_device<Scalar4>(force.data, m_pdata->getN());
}

std::vector<Scalar4> TensorflowComputeAMD::getPositionsArray() const {
    const unsigned int n = m_batch_size == 0 ? m_pdata->getN() : std::min(m_batch_size, m_pdata->getN());
    struct F4 { float x, y, z, w; };
    std::vector<F4> raw = from_device<F4>(htf_get_positions_buffer(m_ctx), n); // fp32 side buffer
    std::vector<Scalar4> out(n);
    for (unsigned int i = 0; i < n; ++i) out[i] = Scalar4{Scalar(raw[i].x), Scalar(raw[i].y), Scalar(raw[i].z), Scalar(raw[i].w)};
    return out;
}

std::vector<Scalar4> TensorflowComputeAMD::getNlistArray() const {
    const size_t n = (size_t)(m_batch_size == 0 ? m_pdata->getN() : std::min(m_batch_size, m_pdata->getN())) * m_nneighs;
    struct F4 { float x, y, z, w; };
    std::vector<F4> raw = from_device<F4>(htf_get_nlist_buffer(m_ctx), n);
    std::vector<Scalar4> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = Scalar4{Scalar(raw[i].x), Scalar(raw[i].y), Scalar(raw[i].z), Scalar(raw[i].w)};
    return out;
}

std::vector<Scalar3> TensorflowComputeAMD::getBoxArray() const {
    return {Scalar3{Scalar(m_box[0]), Scalar(m_box[1]), Scalar(m_box[2])}, Scalar3{Scalar(m_box[3]), Scalar(m_box[4]), Scalar(m_box[5])},
            Scalar3{Scalar(m_box[6]), Scalar(m_box[7]), Scalar(m_box[8])}};
}

std::vector<Scalar> TensorflowComputeAMD::getVirialArray() const {
    const size_t n = (size_t)(m_batch_size == 0 ? m_pdata->getN() : std::min(m_batch_size, m_pdata->getN())) * 9;
    return from_device<Scalar>(htf_get_virial_buffer(m_ctx), n);
}

} // namespace hoomd_tf_amd
