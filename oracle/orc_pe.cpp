// oracle/orc_pe.cpp -- TEST INFRASTRUCTURE: paired-end restatement (placeholder until §8 a17).
#include "orc_internal.h"
extern "C" int orc_search_pe(const orc_index*, const orc_params*, const char*, const char*,
                             const char*, const char*, int64_t*)
{
    return -100;   // not restated yet; the tests that need it are skipped
}
