// FASTA/FASTQ(.gz) reader with klib kseq.h semantics (the reference reads reads through
// kseq_read at /root/reference/src/solver.cpp:230-245): name = header up to the first
// whitespace; sequence = concatenation of the following lines until a line starting with
// '>', '@' or '+'; for FASTQ the quality block is skipped by length.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace dg {
// Appends (name, sequence) pairs. Returns false (err set) if the file cannot be opened.
bool read_sequences(const std::string &path, std::vector<std::pair<std::string, std::string>> &out,
                    std::string &err);
}  // namespace dg
