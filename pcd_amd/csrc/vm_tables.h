// Pointers to one curve's program tables of the wave-per-pairing VM (pairing_vm.hip.h) -- a header of its own so that the context
// (common.h) can hold them without every translation unit depending on the interpreter.
#pragma once
#include <stdint.h>

namespace pcd {

struct VmTables {  // device (or, in the host harness, host) copies of one curve's generated tables
  const uint32_t* progs;   // [nprogs][3]  first step, steps, mask of the state slots written
  const uint32_t* steps;   // [nsteps][3]  kind, first instruction, instructions
  const uint32_t* code;    // 12 words per instruction
  const uint32_t* consts;  // [NCONST][N]
  uint32_t nprogs, nsteps, ncode;  // entries of progs / steps, WORDS of code
};

}  // namespace pcd
