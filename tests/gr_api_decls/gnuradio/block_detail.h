// declarations only (tests/gr_api_decls/README.md): gnuradio/block_detail.h of GNU Radio 3.10 as we read it
#pragma once
#include <gnuradio/buffer_type.h>

#include <memory>
namespace gr {
class block_detail
{
public:
  int ninputs() const;
  int noutputs() const;
  buffer_sptr output(unsigned int which);
};
typedef std::shared_ptr<block_detail> block_detail_sptr;
} // namespace gr
