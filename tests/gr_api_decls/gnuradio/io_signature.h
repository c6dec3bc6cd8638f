// declarations only (tests/gr_api_decls/README.md): gr::io_signature of GNU Radio 3.10
#pragma once
#include <gnuradio/buffer_type.h>

#include <memory>
namespace gr {
class io_signature
{
public:
  typedef std::shared_ptr<io_signature> sptr;
  static sptr make(int min_streams, int max_streams, int sizeof_stream_item);
  static sptr make(int min_streams, int max_streams, int sizeof_stream_item, buffer_type buftype);     // 3.10: custom buffers
  buffer_type stream_buffer_type(size_t index) const;
};
} // namespace gr
