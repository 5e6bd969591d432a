/* The reference's existing C ABI (cloudini_lib/include/cloudini_lib/wasm_functions.h:30-93), served by the MI355X
 * build. Same names, argument meaning and "size on success, 0 on failure" convention; failures additionally leave
 * a message in cldn_LastError(). Buffers are caller-allocated; pointers travel as uintptr_t as in the original
 * (it was designed for WASM linear memory) -- here they are ordinary host addresses. */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

uint32_t cldn_GetHeaderAsYAML(uintptr_t encoded_data_ptr, uint32_t encoded_data_size, uintptr_t output_yaml_ptr);
uint32_t cldn_GetHeaderAsYAMLFromDDS(uintptr_t raw_dds_msg, uint32_t dds_msg_size, uintptr_t output_yaml_ptr);
uint32_t cldn_ComputeCompressedSize(uintptr_t dds_msg_ptr, uint32_t dds_msg_size, float resolution);
uint32_t cldn_GetDecompressedSize(uintptr_t encoded_msg_ptr, uint32_t encoded_msg_size);
uint32_t cldn_ConvertCompressedMsgToPointCloud2Msg(uintptr_t compressed_msg_ptr, uint32_t encoded_data_size,
                                                   uintptr_t output_msg_ptr);
uint32_t cldn_DecodeCompressedData(uintptr_t encoded_data_ptr, uint32_t encoded_data_size, uintptr_t output_data);
uint32_t cldn_DecodeCompressedMessage(uintptr_t compressed_msg_ptr, uint32_t msg_size, uintptr_t output_data_ptr);
uint32_t cldn_EncodePointcloudMessage(const uintptr_t pointcloud_msg_ptr, uint32_t msg_size, float resolution,
                                      uintptr_t output_data_ptr);
uint32_t cldn_EncodePointcloudData(const char* header_as_yaml, const uintptr_t pc_data_ptr, uint32_t pc_data_size,
                                   uintptr_t output_data_ptr);

/* addition: message of the last failure on this thread ("" if none) */
const char* cldn_LastError(void);

#ifdef __cplusplus
}
#endif
