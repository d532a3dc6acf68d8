// Drop-in for the reference's include/ld.h: a client that writes `#include "ld.h"` and uses
// tomahawk::twk_ld / tomahawk::twk_ld_settings (include/ld.h:40-69, include/core.h:909-924) compiles
// unchanged against this directory and links -ltomahawk (libtomahawk.so here is libtomahawk_amd.so under
// the reference library's name).  The reference header also drags in core.h, twk_reader.h and
// writer.h (include/ld.h:30-31): of their types the output record twk1_two_t and its block twk1_two_block_t
// (core.h:756-834, 851-902) are declared here too (twk_two_types.h); the reader / writer classes are not - see
// INTEGRATION.md section 2 for what a client that uses two_reader does.
#ifndef TWK_LD_SHIM_H_
#define TWK_LD_SHIM_H_
#include "twk_ld.h"
#include "twk_two_types.h"
#endif
