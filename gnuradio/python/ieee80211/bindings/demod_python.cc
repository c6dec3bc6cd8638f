/* pybind11 binding of gr::ieee80211::demod: Python name and constructor as in the reference
 * (python/ieee80211/bindings/demod_python.cc:30-43): ieee80211.demod(mupos, mugid). */
#include <pybind11/complex.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace py = pybind11;

#include <gnuradio/ieee80211/demod.h>

void bind_demod(py::module& m)
{
    using demod = ::gr::ieee80211::demod;
    py::class_<demod, gr::block, gr::basic_block, std::shared_ptr<demod>>(
        m, "demod", "VHT / HT / legacy OFDM demodulator (1 antenna): tagged samples in, tagged LLRs out; runs on the MI355X engine")
        .def(py::init(&demod::make), py::arg("mupos"), py::arg("mugid"), "mupos: MU-MIMO user position, mugid: MU-MIMO group id");
}
