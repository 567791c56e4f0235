// A FAKE HOOMD-blue 2.x -- the handful of classes the shim names, with just enough behaviour behind them that
// integration/hoomd_shim/ can be COMPILED (tests/test_shim_compiles.py) and DRIVEN on a GPU
// (tests/test_gpu_shim.py): device-backed GlobalArray / ArrayHandle, a ParticleData and a NeighborList whose
// arrays a test fills (or points at device memory it owns), a ForceCompute base that allocates m_force / m_virial
// the way HOOMD does, a working MaxParticleNumberChange signal.  Signatures follow HOOMD-blue v2.9
// (hoomd/ForceCompute.h, ParticleData.h, BoxDim.h, GlobalArray.h, HalfStepHook.h, md/NeighborList.h).
// This is NOT HOOMD: no integrator, no cell list, no MPI.  A real build points the include path at HOOMD instead
// and never sees this directory.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h> // uchar3: HOOMD takes it from the CUDA / HIP vector types too

#include <cstdint>
#include <functional>
#include <iostream>
#include <memory>
#include <stdexcept>
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

inline void fake_hoomd_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("fake HOOMD: ") + what + ": " + hipGetErrorString(e));
}

//! Device-resident array.  Either owns its memory (resize) or points at memory the test owns (adopt: e.g. a torch
//! tensor's data_ptr -- HOOMD's own arrays are device allocations the plugin never owns either).
template <class T>
class GlobalArray {
public:
    GlobalArray() {}
    GlobalArray(size_t n, unsigned int pitch = 0) { resize(n, pitch); }
    GlobalArray(const GlobalArray &) = delete;
    GlobalArray &operator=(const GlobalArray &) = delete;
    ~GlobalArray() { release(); }
    unsigned int getPitch() const { return m_pitch; }
    size_t getNumElements() const { return m_n; }
    void resize(size_t n, unsigned int pitch = 0) {
        release();
        m_n = n;
        m_pitch = pitch;
        if (n) {
            fake_hoomd_hip(hipMalloc((void **)&m_data, n * sizeof(T)), "hipMalloc(GlobalArray)");
            fake_hoomd_hip(hipMemset(m_data, 0, n * sizeof(T)), "hipMemset(GlobalArray)");
            m_owned = true;
        }
    }
    void adopt(void *device_ptr, size_t n, unsigned int pitch = 0) {
        release();
        m_data = static_cast<T *>(device_ptr);
        m_n = n;
        m_pitch = pitch;
        m_owned = false;
    }
    T *m_data = nullptr;
    size_t m_n = 0;
    unsigned int m_pitch = 0;

private:
    void release() {
        if (m_owned && m_data) (void)hipFree(m_data);
        m_data = nullptr;
        m_owned = false;
    }
    bool m_owned = false;
};
template <class T> using GPUArray = GlobalArray<T>;

//! device: the array's pointer.  host: a staging copy, filled at acquisition (unless overwrite) and written back at
//! release (unless read) -- HOOMD's acquire / release protocol, eagerly.
template <class T>
class ArrayHandle {
    std::vector<T> m_host; // (declared before `data`: stage() fills it while `data` is being initialised)

public:
    ArrayHandle(const GlobalArray<T> &a, access_location::Enum loc = access_location::host,
                access_mode::Enum mode = access_mode::readwrite)
        : data(loc == access_location::device ? a.m_data : stage(a, mode)), m_array(a), m_loc(loc), m_mode(mode) {}
    ~ArrayHandle() {
        if (m_loc == access_location::host && m_mode != access_mode::read && m_array.m_n)
            (void)hipMemcpy(m_array.m_data, m_host.data(), m_array.m_n * sizeof(T), hipMemcpyHostToDevice);
    }
    ArrayHandle(const ArrayHandle &) = delete;
    T *const data;

private:
    T *stage(const GlobalArray<T> &a, access_mode::Enum mode) {
        m_host.resize(a.m_n);
        if (mode != access_mode::overwrite && a.m_n)
            fake_hoomd_hip(hipMemcpy(m_host.data(), a.m_data, a.m_n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy(ArrayHandle)");
        return m_host.data();
    }
    const GlobalArray<T> &m_array;
    access_location::Enum m_loc;
    access_mode::Enum m_mode;
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
    template <class T, R (T::*M)(A...)> void connect(T *obj) {
        m_slots.push_back({obj, [obj](A... a) { return (obj->*M)(a...); }});
    }
    template <class T, R (T::*M)(A...)> void disconnect(T *obj) {
        for (size_t i = 0; i < m_slots.size(); ++i)
            if (m_slots[i].first == obj) {
                m_slots.erase(m_slots.begin() + i);
                return;
            }
    }
    void emit(A... a) {
        for (auto &s : m_slots) s.second(a...);
    }

private:
    std::vector<std::pair<void *, std::function<R(A...)>>> m_slots;
};
} // namespace Nano

class Messenger {
public:
    std::ostream &error() const { return std::cerr; }
    std::ostream &warning() const { return std::cerr; }
    std::ostream &notice(unsigned level) const { return level <= 1 ? std::cerr : m_null; }

private:
    struct NullBuf : std::streambuf {
        int overflow(int c) override { return c; }
    };
    mutable NullBuf m_buf;
    mutable std::ostream m_null{&m_buf};
};

class ExecutionConfiguration {
public:
    bool isCUDAEnabled() const { return true; }
    std::shared_ptr<Messenger> msg = std::make_shared<Messenger>();
};

class ParticleData {
public:
    unsigned int getN() const { return m_N; }
    unsigned int getMaxN() const { return m_max_N; }
    unsigned int getNGhosts() const { return m_n_ghost; }
    unsigned int getNGlobal() const { return m_N; }
    const BoxDim &getBox() const { return m_box; }
    const GlobalArray<Scalar4> &getPositions() const { return m_pos; }
    const GlobalArray<Scalar4> &getNetForce() const { return m_net_force; }
    Nano::Signal<void()> &getMaxParticleNumberChangeSignal() { return m_sig; }
    // --- fake-only: what a test drives
    void setN(unsigned int N, unsigned int max_N, unsigned int n_ghost) {
        const bool grew = max_N != m_max_N;
        m_N = N;
        m_max_N = max_N;
        m_n_ghost = n_ghost;
        if (grew) m_sig.emit(); // ParticleData::reallocate -> m_max_nparticles_signal
    }
    BoxDim m_box;
    GlobalArray<Scalar4> m_pos, m_net_force;
    Nano::Signal<void()> m_sig;
    unsigned int m_N = 0, m_max_N = 0, m_n_ghost = 0;
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
    //! hoomd/ForceCompute.cc: m_force[max N], m_virial[6 x pitch] with pitch = max N rounded up; re-made when max N changes
    explicit ForceCompute(std::shared_ptr<SystemDefinition> sysdef)
        : m_sysdef(sysdef), m_pdata(sysdef->getParticleData()), m_exec_conf(std::make_shared<ExecutionConfiguration>()) {
        allocateForceArrays();
        m_pdata->getMaxParticleNumberChangeSignal().connect<ForceCompute, &ForceCompute::allocateForceArrays>(this);
    }
    virtual ~ForceCompute() { m_pdata->getMaxParticleNumberChangeSignal().disconnect<ForceCompute, &ForceCompute::allocateForceArrays>(this); }
    virtual void compute(unsigned int timestep) { computeForces(timestep); }
    virtual std::vector<std::string> getProvidedLogQuantities() { return {}; }
    virtual Scalar getLogValue(const std::string &, unsigned int) { return Scalar(0); }
    //! hoomd/ForceCompute.cc calcEnergySum: the sum of the w column of m_force over the local particles
    Scalar calcEnergySum() {
        ArrayHandle<Scalar4> f(m_force, access_location::host, access_mode::read);
        double e = 0;
        for (unsigned int i = 0; i < m_pdata->getN(); ++i) e += f.data[i].w;
        return Scalar(e);
    }
    const GlobalArray<Scalar4> &getForceArray() const { return m_force; }
    const GlobalArray<Scalar> &getVirialArray() const { return m_virial; }
    void allocateForceArrays() {
        const unsigned int n = m_pdata->getMaxN();
        m_virial_pitch = (n + 15u) / 16u * 16u; // GlobalArray's 2-D pitch: rows padded to a multiple of 16 elements
        m_force.resize(n);
        m_virial.resize((size_t)6 * m_virial_pitch, m_virial_pitch);
    }

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
    //! the real one checks distances and rebuilds; the fake calls whatever the test installed (its own list builder)
    void compute(unsigned int timestep) {
        ++m_n_compute;
        if (m_on_compute) m_on_compute(timestep);
    }
    const GlobalArray<unsigned int> &getNNeighArray() const { return m_n_neigh; }
    const GlobalArray<unsigned int> &getNListArray() const { return m_nlist; }
    const GlobalArray<unsigned int> &getHeadList() const { return m_head_list; }
    storageMode m_mode = half;
    GlobalArray<unsigned int> m_n_neigh, m_nlist, m_head_list;
    std::function<void(unsigned int)> m_on_compute;
    unsigned int m_n_compute = 0;
};
