/* oracle/orc_tables.c — H.266 constant tables for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Same generated data file as the product (ffvvc_amd/csrc/tables.inc, made by tools/gen_tables.py): these are
 * specification constants, not behaviour. */
#include <stdint.h>
#define VVC355_TABLE(type, name, count) __attribute__((visibility("default"))) const type orc_tab_##name[count]
#include "../ffvvc_amd/csrc/tables.inc"
