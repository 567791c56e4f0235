// pybind11 export: replaces export_TensorflowCompute / export_TensorflowComputeGPU
// (htf/TensorflowCompute.cc:422-486, :617-670) with the same method names, so that
// hoomd/htf/tensorflowcompute.py:136-164 can construct `_htf_amd.TensorflowComputeAMD` where it constructs
// `_htf.TensorflowComputeGPU` today and keep calling hook() / addReferenceForce() / get*Array() unchanged.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "TensorflowComputeAMD.h"

namespace py = pybind11;
using hoomd_tf_amd::FORCE_MODE;
using hoomd_tf_amd::TensorflowComputeAMD;

void hoomd_tf_amd::export_TensorflowComputeAMD(py::module &m) {
    // not exported anywhere else in HOOMD 2.x (.cc:425-426)
    // (HOOMD's integrator calls update() from C++ at the half step; exported so that a driver without one can too)
    py::class_<HalfStepHook, std::shared_ptr<HalfStepHook>>(m, "HalfStepHook").def("update", &HalfStepHook::update);

    py::class_<TensorflowComputeAMD, std::shared_ptr<TensorflowComputeAMD>, ForceCompute>(m, "TensorflowComputeAMD")
        .def(py::init<py::object &, std::shared_ptr<SystemDefinition>, std::shared_ptr<NeighborList>, Scalar, unsigned int,
                      FORCE_MODE, unsigned int, unsigned int>())
        .def("setMappedNlist", &TensorflowComputeAMD::setMappedNlist)
        .def("getPositionsBuffer", &TensorflowComputeAMD::getPositionsBuffer)
        .def("getNlistBuffer", &TensorflowComputeAMD::getNlistBuffer)
        .def("getForcesBuffer", &TensorflowComputeAMD::getForcesBuffer)
        .def("getBoxBuffer", &TensorflowComputeAMD::getBoxBuffer)
        .def("getVirialBuffer", &TensorflowComputeAMD::getVirialBuffer)
        .def("getBatchCapacity", &TensorflowComputeAMD::getBatchCapacity)
        .def("getPositionsArray", &TensorflowComputeAMD::getPositionsArray, py::return_value_policy::take_ownership)
        .def("getNlistArray", &TensorflowComputeAMD::getNlistArray, py::return_value_policy::take_ownership)
        .def("getForcesArray", &TensorflowComputeAMD::getForcesArray, py::return_value_policy::take_ownership)
        .def("getBoxArray", &TensorflowComputeAMD::getBoxArray, py::return_value_policy::take_ownership)
        .def("getVirialArray", &TensorflowComputeAMD::getVirialArray, py::return_value_policy::take_ownership)
        .def("isDoublePrecision", &TensorflowComputeAMD::isDoublePrecision)
        .def("getVirialPitch", &TensorflowComputeAMD::getVirialPitch)
        .def("hook", &TensorflowComputeAMD::getHook)
        .def("addReferenceForce", &TensorflowComputeAMD::addReferenceForce)
        // the lowered model (see TensorflowComputeAMD.h)
        .def("setPotential", &TensorflowComputeAMD::setPotential, py::arg("handle"), py::arg("virial") = false,
             py::arg("check_nlist") = false, py::arg("fused") = 2)
        .def("setTraining", &TensorflowComputeAMD::setTraining);

    py::enum_<FORCE_MODE>(m, "FORCE_MODE").value("tf2hoomd", FORCE_MODE::tf2hoomd).value("hoomd2tf", FORCE_MODE::hoomd2tf);
}

PYBIND11_MODULE(_htf_amd, m) { hoomd_tf_amd::export_TensorflowComputeAMD(m); }
