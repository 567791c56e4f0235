// pybind11 export, replaces export_TensorflowComputeGPU (htf/TensorflowCompute.cc:617-670).
#include <hoomd/extern/pybind/include/pybind11/pybind11.h>
#include "TensorflowComputeAMD.h"

namespace py = pybind11;

PYBIND11_MODULE(_htf_amd, m) {
    py::class_<TensorflowComputeAMD, ForceCompute, std::shared_ptr<TensorflowComputeAMD>>(m, "TensorflowComputeAMD")
        .def("getNlistBuffer", [](TensorflowComputeAMD &c) { return reinterpret_cast<int64_t>(htf_get_nlist_buffer(c.ctx())); })
        .def("getPositionsBuffer", [](TensorflowComputeAMD &c) { return reinterpret_cast<int64_t>(htf_get_positions_buffer(c.ctx())); })
        .def("getVirialBuffer", [](TensorflowComputeAMD &c) { return reinterpret_cast<int64_t>(htf_get_virial_buffer(c.ctx())); })
        .def("isDoublePrecision", [](TensorflowComputeAMD &) { return sizeof(Scalar) == 8; });
}
