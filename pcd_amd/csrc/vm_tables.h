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
  // the same run as ONE list of step records (what the device interpreter walks; progs / steps / script are the host harness's view):
  // [.][4] = kind | largest LIN term count << 8 | flags << 16, first instruction slot, slots, bank flip -- flags 1: last step of a
  // program (apply the flip), 2: select table entry (flags >> 4) first, 4: the field inversion first.  Records 0 .. nflat - 1 are the
  // script; records alone_off .. alone_off + alone_len - 1 the one program the kernel also runs by itself (fe_mul).
  const uint32_t* flat;
  uint32_t nflat, alone_off, alone_len;
};
struct VmCurveTables { VmTables miller, final_exp; };

}  // namespace pcd
