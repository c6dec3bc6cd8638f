// declarations only (tests/gr_api_decls/README.md): the members of gr::basic_block / gr::block that gnuradio/lib/ uses, with the
// access levels and signatures of GNU Radio 3.10 (gnuradio/basic_block.h, gnuradio/block.h, gnuradio/tags.h, gnuradio/sptr_magic.h)
#pragma once
#include <gnuradio/block_detail.h>
#include <gnuradio/io_signature.h>
#include <gnuradio/types.h>
#include <pmt/pmt.h>

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace gr {

struct tag_t {
  uint64_t offset;
  pmt::pmt_t key, value, srcid;
};

class basic_block : public std::enable_shared_from_this<basic_block>
{
public:
  virtual ~basic_block();
  pmt::pmt_t alias_pmt() const;
  void message_port_register_in(pmt::pmt_t port_id);
  void message_port_register_out(pmt::pmt_t port_id);
  void message_port_pub(pmt::pmt_t port_id, pmt::pmt_t msg);
  void _post(pmt::pmt_t which_port, pmt::pmt_t msg);
  template <typename T> void set_msg_handler(pmt::pmt_t which_port, T msg_handler) { (void)which_port; (void)msg_handler; }

protected:
  basic_block(void);
  basic_block(const std::string& name, io_signature::sptr input_signature, io_signature::sptr output_signature);

public:
  io_signature::sptr input_signature() const;
  io_signature::sptr output_signature() const;
};

class block : public basic_block
{
public:
  enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2, TPP_CUSTOM = 3 };
  ~block() override;
  virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required);
  virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                           gr_vector_void_star& output_items);
  virtual bool start();
  virtual bool stop();
  void consume(int which_input, int how_many_items);
  void consume_each(int how_many_items);
  uint64_t nitems_read(unsigned int which_input);
  uint64_t nitems_written(unsigned int which_output);
  void set_tag_propagation_policy(tag_propagation_policy_t p);
  block_detail_sptr detail() const;             // set by the flat flowgraph before the block's thread starts

protected:
  block(void);
  block(const std::string& name, io_signature::sptr input_signature, io_signature::sptr output_signature);
  void add_item_tag(unsigned int which_output, uint64_t abs_offset, const pmt::pmt_t& key, const pmt::pmt_t& value,
                    const pmt::pmt_t& srcid = pmt::PMT_F);
  void get_tags_in_range(std::vector<tag_t>& v, unsigned int which_input, uint64_t abs_start, uint64_t abs_end);
};

} // namespace gr

namespace gnuradio {
template <typename T, typename... Args> std::shared_ptr<T> make_block_sptr(Args&&... args)
{
  return std::shared_ptr<T>(new T(std::forward<Args>(args)...));
}
} // namespace gnuradio
