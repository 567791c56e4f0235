// The fake HOOMD-blue types of hoomd/HOOMDStub.h exported from a module of their own, the way hoomd._hoomd /
// hoomd.md._md export the real ones, so that tests can import the shim module (tests/test_shim_compiles.py) and
// DRIVE it on a GPU (tests/test_gpu_shim.py): the test points the particle data and the neighbor list at device
// arrays it owns (torch tensors), installs its own list builder behind NeighborList::compute, and reads m_force /
// m_virial back.  A fake: nothing here is HOOMD's code or behaviour beyond the signatures the shim uses.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "hoomd/HOOMDStub.h"

namespace py = pybind11;

namespace {
//! a ForceCompute whose m_force a test fills: the "reference force" of hoomd2tf training (addReferenceForce)
class FakeForce : public ForceCompute {
public:
    explicit FakeForce(std::shared_ptr<SystemDefinition> sysdef) : ForceCompute(sysdef) {}
    void setForces(int64_t device_ptr, unsigned int n) {
        fake_hoomd_hip(hipMemcpy(m_force.m_data, reinterpret_cast<const void *>(device_ptr), (size_t)n * sizeof(Scalar4),
                                 hipMemcpyDeviceToDevice), "hipMemcpy(FakeForce)");
    }

protected:
    void computeForces(unsigned int) override {}
};
} // namespace

PYBIND11_MODULE(_hoomd_stub, m) {
    m.attr("scalar_bytes") = (int)sizeof(Scalar);
    // hoomd._hoomd registers these two; the shim's get*Array() return vectors of them (.cc:409-420)
    py::class_<Scalar4>(m, "Scalar4").def_readwrite("x", &Scalar4::x).def_readwrite("y", &Scalar4::y).def_readwrite("z", &Scalar4::z).def_readwrite("w", &Scalar4::w);
    py::class_<Scalar3>(m, "Scalar3").def_readwrite("x", &Scalar3::x).def_readwrite("y", &Scalar3::y).def_readwrite("z", &Scalar3::z);
    py::class_<ParticleData, std::shared_ptr<ParticleData>>(m, "ParticleData")
        .def("getN", &ParticleData::getN)
        .def("getMaxN", &ParticleData::getMaxN)
        .def("setN", &ParticleData::setN, py::arg("N"), py::arg("max_N"), py::arg("n_ghost") = 0)
        .def("setPositionsPtr", [](ParticleData &p, int64_t ptr, size_t n) { p.m_pos.adopt(reinterpret_cast<void *>(ptr), n); })
        .def("setNetForcePtr", [](ParticleData &p, int64_t ptr, size_t n) { p.m_net_force.adopt(reinterpret_cast<void *>(ptr), n); })
        .def("setBox", [](ParticleData &p, std::vector<double> lo, std::vector<double> hi, std::vector<double> tilt,
                          std::vector<int> periodic) {
            p.m_box.m_lo = Scalar3{Scalar(lo[0]), Scalar(lo[1]), Scalar(lo[2])};
            p.m_box.m_hi = Scalar3{Scalar(hi[0]), Scalar(hi[1]), Scalar(hi[2])};
            p.m_box.m_xy = Scalar(tilt[0]);
            p.m_box.m_xz = Scalar(tilt[1]);
            p.m_box.m_yz = Scalar(tilt[2]);
            p.m_box.m_periodic = uchar3{(unsigned char)periodic[0], (unsigned char)periodic[1], (unsigned char)periodic[2]};
        });
    py::class_<SystemDefinition, std::shared_ptr<SystemDefinition>>(m, "SystemDefinition")
        .def(py::init<>())
        .def("getParticleData", &SystemDefinition::getParticleData);
    py::class_<NeighborList, std::shared_ptr<NeighborList>>(m, "NeighborList")
        .def(py::init<>())
        .def("setArrays", [](NeighborList &nl, int64_t n_neigh, int64_t nlist, int64_t head, size_t n, size_t n_entries) {
            nl.m_n_neigh.adopt(reinterpret_cast<void *>(n_neigh), n);
            nl.m_nlist.adopt(reinterpret_cast<void *>(nlist), n_entries);
            nl.m_head_list.adopt(reinterpret_cast<void *>(head), n);
        })
        .def("onCompute", [](NeighborList &nl, std::function<void(unsigned int)> f) { nl.m_on_compute = std::move(f); })
        .def("computeCalls", [](const NeighborList &nl) { return nl.m_n_compute; })
        .def("isFull", [](NeighborList &nl) { return nl.getStorageMode() == NeighborList::full; });
    py::class_<ForceCompute, std::shared_ptr<ForceCompute>>(m, "ForceCompute")
        .def("compute", &ForceCompute::compute)
        .def("calcEnergySum", &ForceCompute::calcEnergySum)
        .def("getLogValue", &ForceCompute::getLogValue)
        .def("getProvidedLogQuantities", &ForceCompute::getProvidedLogQuantities)
        // fake-only: the device addresses of HOOMD's own per-compute arrays, for reading results back
        .def("forcePtr", [](const ForceCompute &f) { return reinterpret_cast<int64_t>(f.getForceArray().m_data); })
        .def("virialPtr", [](const ForceCompute &f) { return reinterpret_cast<int64_t>(f.getVirialArray().m_data); })
        .def("virialPitch", [](const ForceCompute &f) { return f.getVirialArray().getPitch(); })
        .def("forceElements", [](const ForceCompute &f) { return f.getForceArray().getNumElements(); });
    py::class_<FakeForce, std::shared_ptr<FakeForce>, ForceCompute>(m, "FakeForce")
        .def(py::init<std::shared_ptr<SystemDefinition>>())
        .def("setForces", &FakeForce::setForces);
}
