// mkargs.cpp -- DEV / TEST helper of the ISA emulator (tools/emu/gfx950_emu.py): host-side layout of the kernel argument structs and the
// constant block Consts<double> that nmpc_create uploads, from the bytes of an nmpc_config.  Compiled with g++ against the kernel headers
// (their host side); nothing of this is part of the product.
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <vector>
#include "rotors_nmpc.h"
#include "nmpc_consts.hpp"
#include "nmpc_ipm.hpp"
#include "nmpc_team.hpp"
using namespace nmpc;
struct WorkListH { int *count, *done, *list; };
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    nmpc_config g;
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(&g, 1, sizeof(g), f) != sizeof(g)) { fprintf(stderr, "config: expected %zu bytes\n", sizeof(g)); return 3; }
    fclose(f);
    Consts<double> c;
    memset(&c, 0, sizeof(c));
    fill_consts(g, c);
    f = fopen(argv[2], "wb");
    fwrite(&c, 1, sizeof(c), f);
    fclose(f);
#define OFF(S, m) printf("  \"%s.%s\": %zu,\n", #S, #m, offsetof(S, m))
    printf("{\n");
    typedef Work<double> W; typedef Inputs<double> I; typedef Outputs<double> O; typedef TeamWork<double> T; typedef Consts<double> C;
    OFF(W, Bp); OFF(W, AB); OFF(W, bv); OFF(W, qr); OFF(W, xl); OFF(W, ul); OFF(W, LM); OFF(W, iv); OFF(W, iters); OFF(W, status); OFF(W, npol);
    OFF(W, tAB); OFF(W, gbase); OFF(W, prof);
    OFF(I, x0); OFF(I, yref); OFF(I, yref_e); OFF(I, x_init); OFF(I, u_init); OFF(I, yref_bcast);
    OFF(O, u0); OFF(O, x_out); OFF(O, u_out); OFF(O, status);
    OFF(T, tLM); OFF(T, tIV); OFF(T, tP);
    OFF(C, N); OFF(C, steps); OFF(C, polish_ckpt); OFF(C, shared);
    printf("  \"sizeof.Work\": %zu, \"sizeof.Inputs\": %zu, \"sizeof.Outputs\": %zu, \"sizeof.TeamWork\": %zu, \"sizeof.Consts\": %zu, \"sizeof.config\": %zu,\n",
           sizeof(W), sizeof(I), sizeof(O), sizeof(T), sizeof(C), sizeof(g));
    printf("  \"TLM_ROWS\": %d, \"IV_ROWS\": %d, \"TAB_ROWS\": %d, \"TP_ROWS\": %d, \"NX\": %d, \"NU\": %d, \"NY\": %d\n}\n", TLM_ROWS, IV_ROWS, TAB_ROWS, TP_ROWS, NX, NU, NY);
    return 0;
}
