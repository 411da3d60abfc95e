#include "m6a_io.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) { int rc = m6a_io_dataprep(argv[1], argv[2], atoi(argv[3]), 1, 1000, 20, 1, 0, 0); printf("rc %d %s\n", rc, rc ? m6a_io_last_error() : ""); return rc; }
