// HOOMD-blue 2.x side of the drop-in: a ForceCompute that forwards computeForces to
// libhtf_amd.so (include/htf_amd.h).  Goes where htf/TensorflowCompute.{h,cc} are today.
// NOT compiled in this repository's image (no HOOMD headers there); the stand-in driver in
// hoomd_tf_amd/standin.py calls the same entry points in the same order and is what the
// tests and the benchmark exercise.  See INTEGRATION.md.
#pragma once
#include <hoomd/ForceCompute.h>
#include <hoomd/md/NeighborList.h>
#include "htf_amd.h"

class TensorflowComputeAMD : public ForceCompute {
public:
    TensorflowComputeAMD(std::shared_ptr<SystemDefinition> sysdef, std::shared_ptr<NeighborList> nlist,
                         Scalar r_cut, unsigned nneighs, unsigned period, unsigned batch_size,
                         const htf_potential_desc& pot, bool virial, bool check_nlist)
        : ForceCompute(sysdef), m_nlist(nlist) {
        htf_config cfg{};
        cfg.r_cut = r_cut; cfg.nneighs = nneighs; cfg.force_mode = HTF_TF2HOOMD;
        cfg.period = period; cfg.batch_size = batch_size;
        cfg.scalar_dtype = sizeof(Scalar) == 8 ? HTF_F64 : HTF_F32;   // isDoublePrecision()
        cfg.virial = virial; cfg.check_nlist = check_nlist; cfg.max_n = m_pdata->getMaxN();
        check(htf_potential_create(&pot, &m_pot));
        check(htf_create(&cfg, &m_ctx));
        check(htf_set_potential(m_ctx, m_pot));
        if (m_nlist->getStorageMode() == NeighborList::half)          // .cc:74-84
            m_nlist->setStorageMode(NeighborList::full);
        m_pdata->getMaxParticleNumberChangeSignal()
            .connect<TensorflowComputeAMD, &TensorflowComputeAMD::reallocate>(this);
    }
    ~TensorflowComputeAMD() { htf_destroy(m_ctx); htf_potential_destroy(m_pot); }
    htf_ctx* ctx() { return m_ctx; }

protected:
    void computeForces(unsigned int timestep) override {
        m_nlist->compute(timestep);                                    // .cc:162-163
        ArrayHandle<Scalar4> pos(m_pdata->getPositions(), access_location::device, access_mode::read);
        ArrayHandle<unsigned int> n_neigh(m_nlist->getNNeighArray(), access_location::device, access_mode::read);
        ArrayHandle<unsigned int> nl(m_nlist->getNListArray(), access_location::device, access_mode::read);
        ArrayHandle<unsigned int> head(m_nlist->getHeadList(), access_location::device, access_mode::read);
        ArrayHandle<Scalar4> force(m_force, access_location::device, access_mode::overwrite);
        ArrayHandle<Scalar> virial(m_virial, access_location::device, access_mode::readwrite);
        const BoxDim& box = m_pdata->getBox();
        htf_hoomd_arrays a{};
        a.pos = pos.data; a.N = m_pdata->getN(); a.n_ghost = m_pdata->getNGhosts();
        a.n_neigh = n_neigh.data; a.nlist = nl.data; a.head_list = head.data;
        Scalar3 lo = box.getLo(), hi = box.getHi(); uchar3 per = box.getPeriodic();
        a.box = {{lo.x, lo.y, lo.z}, {hi.x, hi.y, hi.z},
                 {box.getTiltFactorXY(), box.getTiltFactorXZ(), box.getTiltFactorYZ()},
                 {per.x, per.y, per.z}};
        a.force = force.data; a.virial = virial.data; a.virial_pitch = m_virial.getPitch();
        check(htf_compute_forces(m_ctx, timestep, &a, /*stream=*/nullptr)); // HOOMD 2.x: default stream
    }
    void reallocate() { check(htf_resize(m_ctx, m_pdata->getMaxN())); }
    void check(int rc) {
        if (rc == HTF_OK) return;
        m_exec_conf->msg->error() << "htf_amd: " << htf_last_error() << std::endl;
        throw std::runtime_error(htf_last_error());                    // .cc:158-159 behaviour
    }
    std::shared_ptr<NeighborList> m_nlist;
    htf_ctx* m_ctx = nullptr;
    htf_potential* m_pot = nullptr;
};
