// declarations only (tests/gr_api_decls/README.md): gnuradio/buffer.h of GNU Radio 3.10 as we read it (gr::buffer::nreaders / reader)
#pragma once
#include <gnuradio/buffer_single_mapped.h>
