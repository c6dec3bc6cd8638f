// declarations only (tests/gr_api_decls/README.md): gnuradio/buffer_type.h of GNU Radio 3.10 as we read it
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
namespace gr {
class block;
class buffer;
typedef std::shared_ptr<block> block_sptr;
typedef std::shared_ptr<buffer> buffer_sptr;
typedef buffer_sptr (*factory_func_ptr)(int nitems, size_t sizeof_item, uint64_t downstream_lcm_nitems, uint32_t downstream_max_out_mult,
                                        block_sptr link, block_sptr buf_owner);
class buffer_type_base
{
public:
  virtual ~buffer_type_base() {}
  bool operator==(const buffer_type_base& other) const { return d_value == other.d_value; }
  bool operator!=(const buffer_type_base& other) const { return d_value != other.d_value; }
  const std::string& name() const { return d_name; }

protected:
  buffer_type_base(const std::string& name, factory_func_ptr factory) : d_name(name), d_factory(factory), d_value(0) {}
  std::string d_name;
  factory_func_ptr d_factory;
  unsigned d_value;
};
typedef const buffer_type_base& buffer_type;
#define MAKE_CUSTOM_BUFFER_TYPE(CLASSNAME, FACTORY_FUNC_PTR)                              \
  class buftype_##CLASSNAME : public gr::buffer_type_base                                 \
  {                                                                                       \
  public:                                                                                 \
    buftype_##CLASSNAME() : gr::buffer_type_base(#CLASSNAME, FACTORY_FUNC_PTR) {}         \
  };
} // namespace gr
