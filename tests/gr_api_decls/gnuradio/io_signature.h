// declarations only (tests/gr_api_decls/README.md): gr::io_signature of GNU Radio 3.10
#pragma once
#include <memory>
namespace gr {
class io_signature
{
public:
  typedef std::shared_ptr<io_signature> sptr;
  static sptr make(int min_streams, int max_streams, int sizeof_stream_item);
};
} // namespace gr
