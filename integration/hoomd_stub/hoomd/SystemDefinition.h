#include "HOOMDStub.h"
