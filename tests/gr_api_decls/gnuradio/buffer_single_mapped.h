// declarations only (tests/gr_api_decls/README.md): gnuradio/buffer.h + gnuradio/buffer_single_mapped.h of GNU Radio 3.10 as we read them
#pragma once
#include <gnuradio/buffer_type.h>

#include <cstddef>
#include <functional>
#include <memory>
namespace gr {
typedef std::function<void*(void*, const void*, std::size_t)> mem_func_t;
class buffer_reader                            // gnuradio/buffer_reader.h
{
public:
  block_sptr link() const;                     // the block that reads via this reader
};
class buffer
{
public:
  virtual ~buffer();
  virtual void post_work(int nitems) = 0;
  virtual void* write_pointer() = 0;
  virtual const void* _read_pointer(unsigned int read_index) = 0;
  bool allocate_buffer(int nitems);
  size_t nreaders() const;                       // gnuradio/buffer.h: the readers attached to this buffer
  buffer_reader* reader(size_t index);

protected:
  buffer(int nitems, size_t sizeof_item, uint64_t downstream_lcm_nitems, uint32_t downstream_max_out_mult, block_sptr link);
  virtual bool do_allocate_buffer(size_t final_nitems, size_t sizeof_item) = 0;
  char* d_base;
  unsigned int d_bufsize;
  size_t d_sizeof_item;
  unsigned int d_write_index;
};
class buffer_single_mapped : public buffer
{
public:
  ~buffer_single_mapped() override;
  virtual bool input_blocked_callback(int items_required, int items_avail, unsigned read_index) = 0;
  virtual bool output_blocked_callback(int output_multiple, bool force = false) = 0;

protected:
  buffer_single_mapped(int nitems, size_t sizeof_item, uint64_t downstream_lcm_nitems, uint32_t downstream_max_out_mult, block_sptr link,
                       block_sptr buf_owner);
  bool do_allocate_buffer(size_t final_nitems, size_t sizeof_item) override = 0;
  bool input_blocked_callback_logic(int items_required, int items_avail, unsigned read_index, char* buffer_ptr, mem_func_t const& memcpy_func,
                                    mem_func_t const& memmove_func);
  bool output_blocked_callback_logic(int output_multiple, bool force, char* buffer_ptr, mem_func_t const& memmove_func);
};
} // namespace gr
