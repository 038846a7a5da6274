// Pointers to one curve's program tables of the wave-per-pairing VM (pairing_vm.hip.h) -- a header of its own so that the context
// (common.h) can hold them without every translation unit depending on the interpreter.
#pragma once
#include <stdint.h>

namespace pcd {

struct VmTables {  // device (or, in the host harness, host) copies of ONE kernel's tables of one curve (Miller loop / final exponentiation)
  const uint32_t* progs;   // [nprogs][3]  first step, steps, mask of the state slots written
  const uint32_t* steps;   // [nsteps][3]  kind | largest LIN term count << 8, first instruction slot, slots
  const uint32_t* code;    // 12 words per instruction slot
  const uint32_t* consts;  // [NCONST][N]
  const uint32_t* script;  // program ids in running order, four per word (0xF0 + i: select table entry i)
  uint32_t nprogs, nsteps, ncode, script_len;  // entries of progs / steps, WORDS of code, entries of script
};
struct VmCurveTables { VmTables miller, final_exp; };

}  // namespace pcd
