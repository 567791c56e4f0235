// A ~100-line stand-in for the HOOMD-blue 2.x headers the shim names -- ONLY so that
// integration/hoomd_shim/ can be compiled (syntax, types, overload resolution, pybind signatures) in an
// image without HOOMD-blue: tests/test_shim_compiles.py.  Signatures follow HOOMD-blue v2.9
// (hoomd/ForceCompute.h, ParticleData.h, BoxDim.h, GlobalArray.h, HalfStepHook.h, md/NeighborList.h); nothing
// here executes.  A real build points the include path at HOOMD instead and never sees this directory.
#pragma once
#include <hip/hip_vector_types.h> // uchar3: HOOMD takes it from the CUDA / HIP vector types too

#include <cstdint>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#ifdef SINGLE_PRECISION
typedef float Scalar;
struct Scalar3 { float x, y, z; };
struct Scalar4 { float x, y, z, w; };
#else
typedef double Scalar;
struct Scalar3 { double x, y, z; };
struct Scalar4 { double x, y, z, w; };
#endif

struct access_location { enum Enum { host, device }; };
struct access_mode { enum Enum { read, readwrite, overwrite }; };

template <class T>
class GlobalArray {
public:
    GlobalArray() {}
    unsigned int getPitch() const { return m_pitch; }
    size_t getNumElements() const { return m_n; }
    T *m_data = nullptr;
    size_t m_n = 0;
    unsigned int m_pitch = 0;
};
template <class T> using GPUArray = GlobalArray<T>;

template <class T>
class ArrayHandle {
public:
    ArrayHandle(const GlobalArray<T> &a, access_location::Enum = access_location::host, access_mode::Enum = access_mode::readwrite)
        : data(a.m_data) {}
    T *const data;
};

class BoxDim {
public:
    Scalar3 getLo() const { return m_lo; }
    Scalar3 getHi() const { return m_hi; }
    Scalar getTiltFactorXY() const { return m_xy; }
    Scalar getTiltFactorXZ() const { return m_xz; }
    Scalar getTiltFactorYZ() const { return m_yz; }
    uchar3 getPeriodic() const { return m_periodic; }
    Scalar3 m_lo{}, m_hi{};
    Scalar m_xy = 0, m_xz = 0, m_yz = 0;
    uchar3 m_periodic{1, 1, 1};
};

namespace Nano {
template <class Sig> class Signal;
template <class R, class... A>
class Signal<R(A...)> {
public:
    template <class T, R (T::*M)(A...)> void connect(T *) {}
    template <class T, R (T::*M)(A...)> void disconnect(T *) {}
};
} // namespace Nano

class Messenger {
public:
    std::ostream &error() const { return std::cerr; }
    std::ostream &warning() const { return std::cerr; }
    std::ostream &notice(unsigned) const { return std::cerr; }
};

class ExecutionConfiguration {
public:
    bool isCUDAEnabled() const { return true; }
    std::shared_ptr<Messenger> msg = std::make_shared<Messenger>();
};

class ParticleData {
public:
    unsigned int getN() const { return 0; }
    unsigned int getMaxN() const { return 0; }
    unsigned int getNGhosts() const { return 0; }
    unsigned int getNGlobal() const { return 0; }
    const BoxDim &getBox() const { return m_box; }
    const GlobalArray<Scalar4> &getPositions() const { return m_pos; }
    const GlobalArray<Scalar4> &getNetForce() const { return m_net_force; }
    Nano::Signal<void()> &getMaxParticleNumberChangeSignal() { return m_sig; }
    BoxDim m_box;
    GlobalArray<Scalar4> m_pos, m_net_force;
    Nano::Signal<void()> m_sig;
};

class SystemDefinition {
public:
    std::shared_ptr<ParticleData> getParticleData() { return m_pdata; }
    std::shared_ptr<ParticleData> m_pdata = std::make_shared<ParticleData>();
};

class Profiler {
public:
    void push(const std::string &) {}
    void push(std::shared_ptr<const ExecutionConfiguration>, const std::string &) {}
    void pop() {}
    void pop(std::shared_ptr<const ExecutionConfiguration>) {}
};

class ForceCompute {
public:
    explicit ForceCompute(std::shared_ptr<SystemDefinition> sysdef)
        : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_exec_conf(std::make_shared<ExecutionConfiguration>()) {}
    virtual ~ForceCompute() {}
    virtual void compute(unsigned int timestep) { computeForces(timestep); }
    virtual std::vector<std::string> getProvidedLogQuantities() { return {}; }
    virtual Scalar getLogValue(const std::string &, unsigned int) { return Scalar(0); }
    Scalar calcEnergySum() { return Scalar(0); }
    const GlobalArray<Scalar4> &getForceArray() const { return m_force; }
    const GlobalArray<Scalar> &getVirialArray() const { return m_virial; }

protected:
    virtual void computeForces(unsigned int timestep) = 0;
    std::shared_ptr<SystemDefinition> m_sysdef;
    std::shared_ptr<ParticleData> m_pdata;
    std::shared_ptr<ExecutionConfiguration> m_exec_conf;
    std::shared_ptr<Profiler> m_prof;
    GlobalArray<Scalar4> m_force;
    GlobalArray<Scalar> m_virial;
    unsigned int m_virial_pitch = 0;
};

class HalfStepHook {
public:
    virtual ~HalfStepHook() {}
    virtual void setSystemDefinition(std::shared_ptr<SystemDefinition> sysdef) = 0;
    virtual void update(unsigned int timestep) = 0;
};

class NeighborList {
public:
    enum storageMode { half, full };
    virtual ~NeighborList() {}
    storageMode getStorageMode() { return m_mode; }
    void setStorageMode(storageMode m) { m_mode = m; }
    void compute(unsigned int) {}
    const GlobalArray<unsigned int> &getNNeighArray() const { return m_n_neigh; }
    const GlobalArray<unsigned int> &getNListArray() const { return m_nlist; }
    const GlobalArray<unsigned int> &getHeadList() const { return m_head_list; }
    storageMode m_mode = half;
    GlobalArray<unsigned int> m_n_neigh, m_nlist, m_head_list;
};
