// TEST INFRASTRUCTURE ONLY (oracle/): empty stand-in for <Pothos/Config.hpp>.
// The reference's ChirpGenerator.hpp:5 includes it but uses nothing from it.
#pragma once
