--  pin_lzma_defect.adb -- test infrastructure (oracle/pin_with_gnat.sh builds it against a checkout of zertovitch/zip-ada where GNAT exists).
--  Encodes a file with the reference's own LZMA.Encoding.Encode at Level_3 and a dictionary SMALLER than the data, the regime in which the
--  BT4 matcher reads positions behind pending bytes that no window fill took up (lz77.adb:1000-1017, 1262-1290; DESIGN.md 10): the stream it
--  writes is compared with tests/golden/lzma_defect.json, where the CPU restatement's stream for the same call is recorded -- together with
--  the fact that it does not decode to the input.
--
--     pin_lzma_defect <input file> <output file> <dictionary size>

with Ada.Command_Line, Ada.Streams.Stream_IO;
with LZMA.Encoding;

procedure Pin_LZMA_Defect is
  use Ada.Streams, Ada.Streams.Stream_IO;
  f_in, f_out : File_Type;
  buf_in : Stream_Element_Array (1 .. 1);
  last : Stream_Element_Offset;

  function More_Bytes return Boolean is
  begin
    return not End_Of_File (f_in);
  end More_Bytes;

  function Read_Byte return LZMA.Byte is
  begin
    Read (f_in, buf_in, last);
    return LZMA.Byte (buf_in (1));
  end Read_Byte;

  procedure Write_Byte (b : LZMA.Byte) is
  begin
    Write (f_out, (1 => Stream_Element (b)));
  end Write_Byte;

  procedure Encode is new LZMA.Encoding.Encode (Read_Byte, More_Bytes, Write_Byte);

begin
  Open (f_in, In_File, Ada.Command_Line.Argument (1));
  Create (f_out, Out_File, Ada.Command_Line.Argument (2));
  Encode
    (level                  => LZMA.Encoding.Level_3,
     literal_context_bits   => 3,
     literal_position_bits  => 0,
     position_bits          => 2,
     end_marker             => True,
     uncompressed_size_info => False,
     dictionary_size        => Natural'Value (Ada.Command_Line.Argument (3)));
  Close (f_in);
  Close (f_out);
end Pin_LZMA_Defect;
