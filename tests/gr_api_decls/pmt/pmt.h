// declarations only (tests/gr_api_decls/README.md): the pmt functions gnuradio/lib/ uses, GNU Radio 3.10 signatures
#pragma once
#include <complex>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
namespace pmt {
class pmt_base;
typedef std::shared_ptr<pmt_base> pmt_t;
extern const pmt_t PMT_T, PMT_F, PMT_NIL;
pmt_t mp(const std::string& s);
pmt_t string_to_symbol(const std::string& s);
pmt_t intern(const std::string& s);
const std::string symbol_to_string(const pmt_t& sym);
bool is_integer(pmt_t x);
bool is_real(pmt_t x);
bool is_c32vector(pmt_t x);
pmt_t from_long(long x);
long to_long(pmt_t x);
pmt_t from_float(double x);
pmt_t from_double(double x);
double to_double(pmt_t x);
const std::vector<std::complex<float>> c32vector_elements(pmt_t v);
pmt_t init_c32vector(size_t k, const std::vector<std::complex<float>>& data);
pmt_t make_dict();
pmt_t dict_add(const pmt_t& dict, const pmt_t& key, const pmt_t& value);
pmt_t make_blob(const void* buf, size_t len);
pmt_t cons(const pmt_t& x, const pmt_t& y);
} // namespace pmt
