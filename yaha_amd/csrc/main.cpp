// main.cpp -- the `yaha` executable: a thin wrapper over the C-ABI library (include/yaha_hip.h).
#include "../../include/yaha_hip.h"
int main(int argc, char **argv) { return yaha_main(argc, argv); }
