// declarations only (tests/gr_api_decls/README.md): gnuradio/types.h of GNU Radio 3.10
#pragma once
#include <complex>
#include <vector>
typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
