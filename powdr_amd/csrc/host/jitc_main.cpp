// powdr_jitc: compiles translation units of run-time specialised kernels in a process of its own.
//
// hiprtc serialises hiprtcCompileProgram calls inside one process (measured on ROCm 7.2: 32 compiler threads, user time =
// wall time), so libpowdr_gpu spreads the units of a prover over several of these helper processes instead (jit.cpp
// compile_all). usage: powdr_jitc <source file> <code object file> [<source> <code object> ...]; exit code 0 = every unit
// compiled; otherwise the first error is written to "<code object file>.err".
#include "../jit.hpp"

#include <cstdio>
#include <fstream>
#include <sstream>

int main(int argc, char** argv) {
    if (argc < 3 || (argc - 1) % 2) { fprintf(stderr, "usage: powdr_jitc <src> <out> [...]\n"); return 2; }
    for (int i = 1; i + 1 < argc; i += 2) {
        std::ifstream in(argv[i], std::ios::binary);
        std::stringstream ss;
        ss << in.rdbuf();
        std::string err;
        std::vector<char> code;
        if (!in || !pw::jit::compile_to_code_object(ss.str(), code, &err)) {
            std::ofstream(std::string(argv[i + 1]) + ".err") << (in ? err : std::string("cannot read the source file"));
            return 1;
        }
        std::ofstream out(argv[i + 1], std::ios::binary);
        out.write(code.data(), (std::streamsize)code.size());
        if (!out) return 1;
    }
    return 0;
}
