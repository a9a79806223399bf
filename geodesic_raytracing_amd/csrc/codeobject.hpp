// codeobject.hpp — kernel source -> gfx950 assembly -> (pass over the compiled code) -> code object.
//
// hiprtc hands back a finished code object.  The fused trace kernel wants one more step between the compiler and the assembler:
// a pass over the register-allocated, scheduled instruction stream (codeobject.cpp: break_vector_runs).  So the program is built
// the way hiprtc builds it - the same clang invocation through the code-object manager (libamd_comgr), with hiprtc's own built-in
// header - but stopped at assembly text, which is then rewritten, assembled and linked in-process.
#pragma once
#include <string>
#include <vector>

namespace gr {

// Source + clang options (what hiprtcCompileProgram would be given) -> assembly text for gfx950.  false + log on failure
// (also when libamd_comgr / libhiprtc-builtins cannot be loaded: the caller then builds through hiprtc without the pass).
bool compile_to_assembly(const std::string& source, const std::vector<std::string>& options, std::string& assembly, std::string& log);

// Assembly text -> loadable code object (assemble + link).
bool assemble_code_object(const std::string& assembly, std::string& code, std::string& log);

struct vector_run_stats {
    int runs_broken = 0;      // vector runs longer than the limit
    int inserted = 0;         // s_nop instructions added
    int longest_before = 0;   // longest run of vector instructions found
    int longest_after = 0;
};

// Inserts `s_nop 0` so that no basic block issues more than `limit` vector-ALU instructions in a row without a scalar one
// (EXPERIMENTS.md C.2: a wave that issues a long pure-vector stretch leaves the SIMD's vector port idle part of the time; one
// scalar instruction per <= 32 vector instructions restores the rate).  A run of L > limit instructions is cut into
// ceil(L / limit) pieces of equal length.  Only lines between two vector instructions are touched, so nothing that must stay
// adjacent (s_getpc_b64 + its offset add, the branch of a waitcnt sequence) is separated.  `only_functions`: restrict the pass to
// these symbols (empty = every function of the file).
vector_run_stats break_vector_runs(std::string& assembly, int limit, const std::vector<std::string>& only_functions = {});

}  // namespace gr
