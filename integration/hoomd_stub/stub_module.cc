// The three HOOMD-blue types the shim's pybind signatures refer to, exported from a module of their own the
// way hoomd._hoomd / hoomd.md._md export the real ones -- only so that tests/test_shim_compiles.py can import
// the shim module.  Nothing here computes.
#include <pybind11/pybind11.h>

#include "hoomd/HOOMDStub.h"

namespace py = pybind11;

PYBIND11_MODULE(_hoomd_stub, m) {
    py::class_<SystemDefinition, std::shared_ptr<SystemDefinition>>(m, "SystemDefinition").def(py::init<>());
    py::class_<NeighborList, std::shared_ptr<NeighborList>>(m, "NeighborList").def(py::init<>());
    py::class_<ForceCompute, std::shared_ptr<ForceCompute>>(m, "ForceCompute");
}
