// STUB (tests/quda_stub/README.md).  Declarations only.
#pragma once
namespace quda {
void getCoords(int x[], int cb_index, const int X[], int parity);
int linkIndexP1(const int x[], const int X[], int mu);
}  // namespace quda
