/* pybind11 binding of gr::ieee80211::demod2: ieee80211.demod2(), as python/ieee80211/bindings/demod2_python.cc of the reference. */
#include <pybind11/complex.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace py = pybind11;

#include <gnuradio/ieee80211/demod2.h>

void bind_demod2(py::module& m)
{
    using demod2 = ::gr::ieee80211::demod2;
    py::class_<demod2, gr::block, gr::basic_block, std::shared_ptr<demod2>>(
        m, "demod2", "HT / VHT 2x2 demodulator (2 antennas): tagged samples in, tagged LLRs out; runs on the MI355X engine")
        .def(py::init(&demod2::make));
}
