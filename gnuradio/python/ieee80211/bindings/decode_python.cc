/* pybind11 binding of gr::ieee80211::decode: ieee80211.decode(ifdebug), as python/ieee80211/bindings/decode_python.cc of the reference. */
#include <pybind11/complex.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace py = pybind11;

#include <gnuradio/ieee80211/decode.h>

void bind_decode(py::module& m)
{
    using decode = ::gr::ieee80211::decode;
    py::class_<decode, gr::block, gr::basic_block, std::shared_ptr<decode>>(
        m, "decode", "soft Viterbi decoder + FCS check: tagged LLRs in, PDUs on message port \"out\"; runs on the MI355X engine")
        .def(py::init(&decode::make), py::arg("ifdebug"), "ifdebug: print the per-packet debug lines");
}
