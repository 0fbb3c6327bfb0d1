// serial_host_probe.cpp -- host-only checks of the save / load layer of include/seal/seal.h (no GPU, no libhefx: nothing
// here touches an engine).  Prints, for tests/test_shim_host_cpu.py to compare with Python's hashlib:
//   sha3 <hex digest of argv[1] as bytes>           (SHA3-256, SEAL's parms_id hash)
//   parms_id <4 x uint64 hex> <57..-byte stream hex> (EncryptionParameters::parms_id and ::Save of a CKKS set)
#include <cstdio>
#include <iostream>
#include <sstream>

#include "seal/seal.h"

using namespace seal;

int main(int argc, char **argv)
{
    const std::string msg = argc > 1 ? argv[1] : "";
    const auto d = shim::sha3_256(reinterpret_cast<const std::uint8_t *>(msg.data()), msg.size());
    std::printf("sha3 ");
    for (auto b : d) std::printf("%02x", b);
    std::printf("\n");
    EncryptionParameters p(scheme_type::CKKS);
    p.set_poly_modulus_degree(8192);
    p.set_coeff_modulus({SmallModulus(0xffffffffffe8001ull), SmallModulus(0xfffff4c001ull), SmallModulus(0xfffffdc001ull),
                         SmallModulus(0xfffffffffffc001ull)});
    const parms_id_type id = p.parms_id();
    std::printf("parms_id %016llx %016llx %016llx %016llx ", (unsigned long long)id[0], (unsigned long long)id[1],
                (unsigned long long)id[2], (unsigned long long)id[3]);
    std::stringstream ss;
    EncryptionParameters::Save(p, ss);
    for (unsigned char c : ss.str()) std::printf("%02x", c);
    std::printf("\n");
    EncryptionParameters q = EncryptionParameters::Load(ss);
    std::printf("roundtrip %d\n", q == p ? 1 : 0);
    return 0;
}
