Route #1: 124
Route #2: 147 101 80
Route #3: 74 59 151
Route #4: 24 38
Route #5: 40 115
Route #6: 109 2
Route #7: 113 49
Route #8: 75 52
Route #9: 117 118
Route #10: 71 148
Route #11: 58 97
Route #12: 150 143
Route #13: 157 47
Route #14: 95 90
Route #15: 125 36
Route #16: 27 162
Route #17: 163 50
Route #18: 100 153
Route #19: 126 133 111
Route #20: 160 46
Route #21: 12 34
Route #22: 84 85
Route #23: 65 137
Route #24: 106 16 32 41
Route #25: 112 122 127 121
Route #26: 91 7 108 164 99 142 107 37 119 105
Route #27: 94 96 89 98 93 158
Route #28: 61 132 161 18 51
Route #29: 152 87 103 155 149 114
Route #30: 22 62 92 110 70 131
Route #31: 60 69 13 171 67
Route #32: 145 146 53 11 68
Route #33: 48 141 138
Route #34: 102 88
Route #35: 28 25
Route #36: 123 144 120
Route #37: 165 168
Route #38: 129 130
Route #39: 159 116
Route #40: 167 35
Route #41: 83 21
Route #42: 26 139
Route #43: 76 55
Route #44: 166 3
Route #45: 57 20 14
Route #46: 54 4 33 17 135
Route #47: 134 156 86
Route #48: 154 1 140 19 104 42
Route #49: 23 10 5 9 29 8 64 31
Route #50: 45 56 39 81 78 82 128
Route #51: 169 77 170 136
Route #52: 66 30 79 44 72
Route #53: 63 73 15 43 6
Cost 45607
