// Host-visible copies of the H.266 constant tables (generated: tables.inc; generator: tools/gen_tables.py).
// Exported so that a host integration (and the tests) index the same data the kernels use, e.g. to pick the
// interpolation filter of a motion-vector fraction the way libavcodec/vvc/vvc_inter.c:187-189 does.
#include <stdint.h>
#define VVC355_TABLE(type, name, count) extern "C" __attribute__((visibility("default"))) const type vvc355_tab_##name[count]
#include "tables.inc"
// the small tables the reference keeps inline in its .c files (tools/gen_tables.py, main_small): the kernels use these same initialisers
// directly (loopfilter.hip, alf.hip) or prove their packed forms equal to them at compile time (intra.hip, itx.hip, alf.hip)
#include "tables_small.inc"
