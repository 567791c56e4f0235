#include "../HOOMDStub.h"
