/* module ieee80211_python: the three replaced blocks; with C8P_REFERENCE_TREE the CMake list adds the reference's own
 * *_python.cc files for every other block and defines C8P_WITH_REFERENCE_BLOCKS so that they are registered too. */
#include <pybind11/pybind11.h>

#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

namespace py = pybind11;

void bind_demod(py::module& m);
void bind_demod2(py::module& m);
void bind_decode(py::module& m);
#ifdef C8P_WITH_REFERENCE_BLOCKS
void bind_trigger(py::module& m);
void bind_sync(py::module& m);
void bind_signal(py::module& m);
void bind_modulation(py::module& m);
void bind_encode(py::module& m);
void bind_signal2(py::module& m);
void bind_pktgen(py::module& m);
void bind_encode2(py::module& m);
void bind_pad(py::module& m);
void bind_modulation2(py::module& m);
void bind_pad2(py::module& m);
#endif

// numpy must be initialised before any binding touches arrays (the usual OOT boilerplate)
void* init_numpy()
{
    import_array();
    return NULL;
}

PYBIND11_MODULE(ieee80211_python, m)
{
    init_numpy();
    py::module::import("gnuradio.gr");      // gr::block / gr::basic_block must be registered first
    bind_demod(m);
    bind_demod2(m);
    bind_decode(m);
#ifdef C8P_WITH_REFERENCE_BLOCKS
    bind_trigger(m);
    bind_sync(m);
    bind_signal(m);
    bind_modulation(m);
    bind_encode(m);
    bind_signal2(m);
    bind_pktgen(m);
    bind_encode2(m);
    bind_pad(m);
    bind_modulation2(m);
    bind_pad2(m);
#endif
}
